"""dev: the verdict's acceptance run for two waves per SIMD: the library built with -DP1_WAVES_PER_SIMD=2 (tools/mkvariant.sh occ2 ...), per-step kernel forced
(ARMOUR_OPT_P1_BUILD = 1), B in {16, 64, 128} x 12 fresh builds, digest of all tables against the shipped library's.
    python tools/gpu_p1_occ2_stress.py"""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import hashlib
    import numpy as np
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    for B in (16, 64, 128):
        bp = random_batch(7, B, 20); ks = random_k(3, B)
        hs, ms = [], []
        for rep in range(int(sys.argv[2])):
            nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            g, jac = nlp.eval_g_jac(ks)
            hs.append(hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes() + np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(jac).tobytes()).hexdigest()[:12])
            ms.append(nlp.build_ms); nlp.close()
        print(f"B={B}: digests {sorted(set(hs))} over {len(hs)} builds; build ms min {min(ms):.2f} median {sorted(ms)[len(ms)//2]:.2f}", flush=True)
    sys.exit(0)
for name, reps in (("head", 3), ("occ2", 12)):
    e = dict(os.environ)
    if name != "head": e["ARMOUR_HIP_LIB"] = f"/root/repo/armour_amd/lib/libarmour_hip_{name}.so"
    r = subprocess.run([sys.executable, __file__, "child", str(reps)], env=e, capture_output=True, text=True, timeout=900)
    print(f"== {name}\n{r.stdout.strip()}\n{r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
