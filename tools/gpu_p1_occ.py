"""dev: P1 build time and a hash of the tables at several batch sizes, one child process per (library, environment) variant --
the harness of profiles/r02_p1_two_waves_per_simd.txt (A/B libraries are built by hand into armour_amd/lib/libarmour_hip_<name>.so)."""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import hashlib
    import numpy as np
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in [int(x) for x in sys.argv[2:]]:
        bp = random_batch(0, B, 20)
        ms = []
        for rep in range(2):
            nlp = ArmourNLP(T=100)
            nlp.set_parameters(bp['q0'], bp['qd0'], bp['qdd0'], bp['q_des'], bp['obstacles'])
            ms.append(nlp.build_ms)
        h = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()).hexdigest()[:12]
        print(f"  B={B}: build ms {['%.3f' % m for m in ms]}  tables {h}", flush=True)
    sys.exit(0)
OCC1 = {"ARMOUR_HIP_LIB": "/root/repo/armour_amd/lib/libarmour_hip_occ1.so"}
W1 = {"ARMOUR_P1_WAVES": "1"}
LIB = lambda n: {"ARMOUR_HIP_LIB": f"/root/repo/armour_amd/lib/libarmour_hip_{n}.so"}
for env in [{}, {"ARMOUR_P1_MAX_WAVES_PER_CU": "2"}]:   # edit: LIB("name") selects armour_amd/lib/libarmour_hip_<name>.so (an A/B build)
    print("env", env, flush=True)
    e = dict(os.environ); e.update(env)
    Bs = sys.argv[1:] or ["1", "16", "128"]
    r = subprocess.run([sys.executable, __file__, "child"] + Bs, env=e, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:], flush=True)
