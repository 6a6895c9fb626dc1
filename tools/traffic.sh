#!/bin/bash
# dev: L2-miss traffic of the reach-set build kernels for one or more libraries:  tools/traffic.sh B lib [lib...]   (lib: 'tree' or the <name> of
# armour_amd/lib/libarmour_hip_<name>.so).  One counter per rocprofv3 pass (FETCH_SIZE, WRITE_SIZE, TCC hit / miss), each under its own timeout;
# a 'library' of the form opt:ID=VALUE[;ID=VALUE] is the tree's library with those per-handle options.
# prints, per library, the mean per dispatch of the chain / tv kernel and 2 x FETCH_SIZE + WRITE_SIZE in bytes (MI355X_MICROARCH.md, HBM section).
set -u
export TMPDIR=/tmp
R=$PWD
B=$1; shift
OUT=$R/gpurun_out/p1_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for lib in "$@"; do
  OPTS=""
  case "$lib" in opt:*) OPTS="$(echo ${lib#opt:} | tr ";" " ")"; unset ARMOUR_HIP_LIB;; tree) unset ARMOUR_HIP_LIB;; *) export ARMOUR_HIP_LIB=$R/armour_amd/lib/libarmour_hip_$lib.so;; esac
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 180 rocprofv3 --pmc $c --output-format csv -d $OUT/${lib}_$n -- python3 $R/tools/workload.py p1 $B $OPTS > $OUT/${lib}_$n.log 2>&1 || echo "pass $lib $n failed"
  done
done
cd $R
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for lib in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(f'gpurun_out/p1_traffic/{lib}_*/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            if 'chain_kernel' in row['Kernel_Name'] or 'tv_kernel' in row['Kernel_Name']:
                a = acc[row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
    m = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        tot = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
        hit = m.get('TCC_HIT_sum', 0) / max(1.0, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0))
        print(f"{lib:>10s}: reads {2 * m['FETCH_SIZE'] * 1024 / 1e9:7.3f} GB  writes {m['WRITE_SIZE'] * 1024 / 1e9:7.3f} GB  total {tot / 1e9:7.3f} GB  L2 hit {hit:.3f}  ({ {k: v[1] for k, v in acc.items()} } dispatches)")
    else:
        print(lib, "incomplete:", m)
PY
