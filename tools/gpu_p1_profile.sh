#!/bin/bash
# dev only: cycle attribution inside the P1 chain kernel (wave 0)
# NOTE: the -DP1_PROFILE build last worked before the role-wave rework; run against the current kernels it aborted with
# HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  A faulting kernel can take the GPU box down: repair the instrumentation
# (suspect: the static prof_lds block next to the dynamic LDS of the 3-wave launch) before running this again.
if [ "${P1_PROFILE_FORCE:-0}" != "1" ]; then echo "gpu_p1_profile.sh: disabled, see the note in the script"; exit 1; fi
make -C armour_amd/csrc -B EXTRA="-DP1_PROFILE" >/dev/null 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'.')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem
p = random_problem(0, 20)
nlp = ArmourNLP(T=100).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
print("build ms", nlp.build_ms)
PY
