"""dev: reach-set build of several libraries (ARMOUR_HIP_LIB) at several batch sizes on ONE box: best build ms of N (interleaved rounds) and a
sha1 of the tables (torque radii, link generators, every link / torque PZ of the first and last problem) -- a value-preserving change of the
build kernels must reproduce the reference library's digests bit for bit.
    python tools/gpu_p1_ab_digest.py <lib name or 'tree'>... [-- B...]        (lib name: armour_amd/lib/libarmour_hip_<name>.so)"""
import hashlib, os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in [int(x) for x in sys.argv[2:]]:
        bp = random_batch(5, B, 20)
        nlp = ArmourNLP(T=100)
        ms = []
        for _ in range(5 if B <= 16 else 3):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); ms.append(nlp.build_ms)
        h = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes())
        for b in sorted({0, B - 1}):
            for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
                for i in range(cnt):
                    for t in range(0, nlp.T, 7):
                        c, ind, keys, co = nlp.pz(which, i, t, b=b)
                        h.update(np.ascontiguousarray(c).tobytes() + np.ascontiguousarray(ind).tobytes() + np.ascontiguousarray(keys).tobytes() + np.ascontiguousarray(co).tobytes())
        info = nlp.build_info()
        print(B, min(ms), h.hexdigest()[:16], info["kernel"], info["waves"], flush=True); nlp.close()
    sys.exit(0)
args = sys.argv[1:]
libs = args[:args.index("--")] if "--" in args else args
Bs = args[args.index("--") + 1:] if "--" in args else ["1", "16", "128"]
best, dig = {}, {}
for rnd in range(3):
    for lib in libs:
        e = dict(os.environ)
        if lib != "tree": e["ARMOUR_HIP_LIB"] = f"/root/repo/armour_amd/lib/libarmour_hip_{lib}.so"
        r = subprocess.run([sys.executable, __file__, "child"] + Bs, env=e, capture_output=True, text=True, timeout=900)
        for line in r.stdout.splitlines():
            f = line.split()
            if len(f) < 5 or not f[0].isdigit(): continue
            B, ms, h = f[0], float(f[1]), f[2]
            best[(lib, B)] = min(best.get((lib, B), 1e9), ms)
            dig.setdefault((lib, B), set()).add(h)
        if r.returncode: print(lib, "FAILED", r.stderr[-400:])
print("B      " + "  ".join(f"{l:>28s}" for l in libs))
for B in Bs: print(f"{B:>5s}  " + "  ".join(f"{best.get((l, B), float('nan')):9.3f} ms {'/'.join(sorted(dig.get((l, B), {'-'}))):>16s}" for l in libs), flush=True)
ok = all(dig.get((l, B)) == dig.get((libs[0], B)) and len(dig.get((l, B), ())) == 1 for l in libs for B in Bs)
print("digests identical across libraries and rounds:", ok)
