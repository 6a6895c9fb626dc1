#!/bin/bash
# dev: the eight-wave time-vectorised blocks (dedicated helper waves, two waves per SIMD) against the four-wave blocks: times, and
# table digests across repeated builds (a build of the library with -DP1_TV_WAVES_PER_SIMD=2 must be in tree)
for ded in 1 0 1 0; do
  echo "dedicated=$ded" $(ARMOUR_P1_TV_DEDICATED=$ded ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep -o "blocks of [0-9]* wave.*arena each), [0-9.]* ms, flags 0x[0-9a-f]*" | sed 's/(.*arena each),//' | tr '\n' ' ')
done
cat > /tmp/ded_digest.py <<'PY'
import hashlib, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k
for B, O in ((24, 20), (64, 20), (128, 20)):
    bp = random_batch(7, B, O); ks = random_k(3, B)
    hs = []
    for rep in range(int(sys.argv[1])):
        nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        g, jac = nlp.eval_g_jac(ks)
        keys = hashlib.sha1(np.ascontiguousarray(nlp.link_generators()).tobytes()).hexdigest()[:10]
        full = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(jac).tobytes()).hexdigest()[:10]
        hs.append((keys, full)); nlp.close()
    print(f"dedicated={os.environ.get('ARMOUR_P1_TV_DEDICATED','1')} B={B}: link generators {sorted(set(h[0] for h in hs))} radii+g+jac {sorted(set(h[1] for h in hs))} over {len(hs)} builds", flush=True)
PY
ARMOUR_P1_TV_DEDICATED=1 timeout -k 10 300 python /tmp/ded_digest.py 6 || exit 1
ARMOUR_P1_TV_DEDICATED=0 timeout -k 10 300 python /tmp/ded_digest.py 2 || exit 1
