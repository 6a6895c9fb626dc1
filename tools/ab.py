"""A/B of the reach-set build on ONE GPU box: best build time of N interleaved rounds and a digest of every table, per variant and batch size.
    python tools/ab.py <variant>... [-- B...] [--rounds R] [--reps N] [--obstacles O] [--seed S]
A variant is  tree | <name>  (armour_amd/lib/libarmour_hip_<name>.so, built by tools/variant.sh), optionally followed by per-handle options:
    tree@116=0,122=64        the shipped library with ARMOUR_OPT 116 = 0 and 122 = 64 (include/armour_hip.h)
    occ2@105=1               the occ2 variant with the per-step kernel forced
The digest (sha1 of torque radii, link generators, the fused g / Jacobian at a seeded k, every 7th link / torque PZ of the first and last problem)
must be the FIRST variant's, bit for bit, in every round: a value-preserving change reproduces it; anything else prints `digests differ`
and exits 1.  `--reps` fresh handles are built per (variant, B, round) -- a stress of launch-to-launch reproducibility when raised."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(opts, Bs, reps, O, seed):
    import numpy as np
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    for B in Bs:
        bp = random_batch(seed, B, O); ks = random_k(3, B)
        ms, hs = [], set()
        for rep in range(reps):
            nlp = ArmourNLP(T=100)
            for kv in opts:
                nlp.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
            for _ in range(2 if rep == 0 else 1):   # (the first build of a process loads the code object)
                nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            ms.append(nlp.build_ms)
            g, jac = nlp.eval_g_jac(ks)
            h = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in (nlp.torque_radius(), nlp.link_generators(), g, jac)))
            if rep == 0:
                for b in sorted({0, B - 1}):
                    for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
                        for i in range(cnt):
                            for t in range(0, nlp.T, 7):
                                h.update(b"".join(np.ascontiguousarray(a).tobytes() for a in nlp.pz(which, i, t, b=b)))
                first = h.hexdigest()[:16]
                h = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in (nlp.torque_radius(), nlp.link_generators(), g, jac)))
            hs.add(h.hexdigest()[:16])
            info = nlp.build_info(); nlp.close()
        print("AB", B, min(ms), first, len(hs), info["kernel"], info["waves"], flush=True)


def main(argv):
    if argv[:1] == ["child"]:
        return child([x for x in argv[1].split(",") if x], [int(b) for b in argv[2].split(",")], int(argv[3]), int(argv[4]), int(argv[5]))
    def flag(name, default):
        if name in argv:
            i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return int(v)
        return default
    rounds, reps, O, seed = flag("--rounds", 3), flag("--reps", 1), flag("--obstacles", 20), flag("--seed", 5)
    variants = argv[:argv.index("--")] if "--" in argv else argv
    Bs = argv[argv.index("--") + 1:] if "--" in argv else ["1", "16", "128"]
    best, dig, unstable, kern = {}, {}, set(), {}
    for rnd in range(rounds):
        for v in variants:
            lib, _, opts = v.partition("@")
            e = dict(os.environ)
            if lib != "tree":
                e["ARMOUR_HIP_LIB"] = os.path.join(ROOT, "armour_amd", "lib", f"libarmour_hip_{lib}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", opts, ",".join(Bs), str(reps), str(O), str(seed)], env=e, capture_output=True, text=True, timeout=1100)
            for line in r.stdout.splitlines():
                f = line.split()
                if f[:1] != ["AB"]:
                    continue
                key = (v, f[1])
                best[key] = min(best.get(key, 1e9), float(f[2])); dig.setdefault(key, set()).add(f[3]); kern[key] = f"{f[5]}x{f[6]}"
                if f[4] != "1":
                    unstable.add(key)
            if r.returncode:
                print(v, "FAILED", r.stderr[-600:], flush=True)
        print(f"round {rnd + 1} of {rounds} done", flush=True)
    w = max(len(v) for v in variants) + 2
    for B in Bs:
        for v in variants:
            k = (v, B)
            print(f"B {B:>4s}  {v:<{w}s} {best.get(k, float('nan')):9.3f} ms  {'/'.join(sorted(dig.get(k, {'-'})))}  {kern.get(k, '')}{'  UNSTABLE within a process' if k in unstable else ''}")
    ok = not unstable and all(dig.get((v, B)) == dig.get((variants[0], B)) and len(dig.get((v, B), ())) == 1 for v in variants for B in Bs)
    print("digests identical across variants, rounds and repeats:" if ok else "digests differ:", ok)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main(sys.argv[1:])
