"""dev: reach-set build ms of several libraries (ARMOUR_HIP_LIB) at several batch sizes, interleaved rounds, best of N per (library, B).
    python tools/gpu_p1_ab_libs.py <lib name or 'tree'>... [-- B...]"""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in [int(x) for x in sys.argv[2:]]:
        bp = random_batch(5, B, 20)
        nlp = ArmourNLP(T=100)
        ms = []
        for _ in range(5 if B <= 16 else 3):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); ms.append(nlp.build_ms)
        print(B, min(ms), flush=True); nlp.close()
    sys.exit(0)
args = sys.argv[1:]
libs = args[:args.index("--")] if "--" in args else args
Bs = args[args.index("--") + 1:] if "--" in args else ["1", "2", "4", "16", "128"]
best = {}
for rnd in range(3):
    for lib in libs:
        e = dict(os.environ)
        if lib != "tree": e["ARMOUR_HIP_LIB"] = f"/root/repo/armour_amd/lib/libarmour_hip_{lib}.so"
        r = subprocess.run([sys.executable, __file__, "child"] + Bs, env=e, capture_output=True, text=True, timeout=900)
        for line in r.stdout.splitlines():
            B, ms = line.split(); best[(lib, B)] = min(best.get((lib, B), 1e9), float(ms))
        if r.returncode: print(lib, "FAILED", r.stderr[-300:])
print("B      " + "  ".join(f"{l:>10s}" for l in libs))
for B in Bs: print(f"{B:>5s}  " + "  ".join(f"{best.get((l, B), float('nan')):10.3f}" for l in libs), flush=True)
