#!/bin/bash
# dev only: static instruction counts between the P2_TIMELINE stamps of the one-point fused kernel
cd /root/repo/armour_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DP2_TIMELINE -S --cuda-device-only p2_eval.hip -o /tmp/p2_tl.s 2>/dev/null
cd /tmp; a=$(grep -n "^_ZN.*ILb1ELb1ELb0EE.*:" p2_tl.s | cut -d: -f1); b=$(grep -n "^_ZN.*ILb1ELb0ELb1EE.*:" p2_tl.s | cut -d: -f1); sed -n "${a},${b}p" p2_tl.s > k_tl.s
python3 - <<'PY'
from collections import Counter
L=open('/tmp/k_tl.s').read().split('\n')
st=[i for i,l in enumerate(L) if 's_memrealtime' in l]
for a,b in zip(st[:-1],st[1:]):
    seg=[l.strip() for l in L[a:b] if l.startswith('\t') and not l.strip().startswith(('.',';'))]
    c=Counter(x.split()[0].split('_')[0] for x in seg)
    print(a,b,len(seg),dict(c), 'branches', sum(1 for x in seg if 'cbranch' in x), 'waitcnt', sum(1 for x in seg if 'waitcnt' in x))
PY
