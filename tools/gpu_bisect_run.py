"""dev: run a historical tree (ab_head/<commit>/, built by tools/bisect_build.sh with two waves per SIMD) on the case that gave wrong tables in round 2:
B = 16 per-step (768 one-wave blocks, three per CU), REPS fresh builds, digest of the tables of each.
    python tools/gpu_bisect_run.py <commit> [<commit> ...]"""
import os, subprocess, sys
if len(sys.argv) > 2 and sys.argv[1] == "child":
    root = sys.argv[2]
    sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
    import hashlib
    import numpy as np
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in (16, 64):
        bp = random_batch(0, B, 20)
        out = []
        for rep in range(6):
            nlp = ArmourNLP(T=100)
            try:
                nlp.set_parameters(bp['q0'], bp['qd0'], bp['qdd0'], bp['q_des'], bp['obstacles'])
                out.append(hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()).hexdigest()[:10] + " %.1fms" % nlp.build_ms)
            except Exception as e:
                out.append("ERR " + str(e)[:50])
            del nlp
        print(f"B={B}: " + " | ".join(out), flush=True)
    sys.exit(0)
for c in sys.argv[1:]:
    root = "/root/repo" if c == "head" else f"/root/repo/ab_head/{c}"
    e = dict(os.environ); e["ARMOUR_P1_TV"] = "0"; e.pop("ARMOUR_HIP_LIB", None)
    r = subprocess.run([sys.executable, __file__, "child", root], env=e, capture_output=True, text=True, timeout=400, cwd=root)
    print(f"== {c}\n{r.stdout.strip()[-1500:]}\n{r.stderr.strip()[-400:] if r.returncode else ''}", flush=True)
