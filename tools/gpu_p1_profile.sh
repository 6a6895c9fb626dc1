#!/bin/bash
# dev only: cycle attribution inside the P1 chain kernel (wave 0)
# (The counters live in each wave's own registers / stack; an earlier version kept them in a static LDS block next to the
# dynamic LDS of the 3-wave launch and aborted with a memory aperture violation.)  Prints a lot: expect ~15 ms builds.
make -C armour_amd/csrc -B EXTRA="-DP1_PROFILE" >/dev/null 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'.')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem
p = random_problem(0, 20)
nlp = ArmourNLP(T=100).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
print("build ms", nlp.build_ms)
PY
