"""dev: the -DP1_STAMPS log of `tools/workload.py p1 1` (last build) as one merged timeline of the main and the helper block of t = T - 1,
plus the per-wave summaries of t = 0 / 60 / T - 1.   python tools/dev/stamps_timeline.py gpurun_out/<log>"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
summ = [l for l in L if re.match(r'\[t=\d+ (helper )?wave \d', l) and 'signal' not in l]
n = len(summ) // 3
for l in sorted(summ[-n:]): print(l)
sig = [l for l in L if 'signal]' in l]
n = len(sig) // 3
names = {0: 'C1', 1: 'C0', 2: 'CC2', 3: 'B1', 4: 'B2', 5: 'CA', 6: 'C3', 7: 'C3P'}
rows = []
for l in sig[-n:]:
    m = re.match(r'\[t=(\d+) (helper )?wave (\d) signal\] word (-?\d+) value (\d+) at (\d+)', l)
    t, h, wv, word, val, at = m.groups()
    word = int(word)
    nm = names.get(word, ('take%d.%d waited(k)' % ((word - 300) // 16, (word - 300) % 16)) if 300 <= word < 1000 else ('X%d.%d' % ((word - 100) // 16, (word - 100) % 16) if 100 <= word < 300 else str(word)))
    rows.append((int(at), ('H' if h else 'M') + wv, nm, int(val)))
rows.sort()
for r in rows: print("%8d %s %s %d" % r)
