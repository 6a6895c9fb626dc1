"""dev: the per-step reach-set kernel built with two waves per SIMD (variants libarmour_hip_<name>.so made by tools/mkvariant.sh with
-DP1_WAVES_PER_SIMD=2 and one more switch each): B = 16 (768 one-wave blocks: three per CU), REPS fresh builds per variant, digest of the
tables of each -- 'stable' means every build gives the digest of the shipped library.
    python tools/gpu_p1_occ2_hunt.py name [name ...]"""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import hashlib
    import numpy as np
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    B = int(sys.argv[2])
    bp = random_batch(0, B, 20)
    out = []
    for rep in range(int(sys.argv[3])):
        nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 1)
        try:
            nlp.set_parameters(bp['q0'], bp['qd0'], bp['qdd0'], bp['q_des'], bp['obstacles'])
            out.append(hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()).hexdigest()[:10] + " %.2fms" % nlp.build_ms)
        except Exception as e:
            out.append("ERR " + str(e)[:60])
        nlp.close()
    print(" | ".join(out), flush=True)
    sys.exit(0)
for name in sys.argv[1:]:
    e = dict(os.environ)
    if name != "head": e["ARMOUR_HIP_LIB"] = f"/root/repo/armour_amd/lib/libarmour_hip_{name}.so"
    for B in (16, 40):
        r = subprocess.run([sys.executable, __file__, "child", str(B), "6"], env=e, capture_output=True, text=True, timeout=300)
        print(f"{name:10s} B={B}: {r.stdout.strip()[-700:]} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
