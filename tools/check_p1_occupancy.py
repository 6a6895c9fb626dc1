#!/usr/bin/env python3
"""Build-time check (armour_amd/csrc/Makefile): every kernel of p1_reach.hip that calls the wave-level PZ operators must compile
to ONE wave per SIMD.  Reads the compiler's -Rpass-analysis=kernel-resource-usage remarks of that object and fails the build when
a chain / pzop / tv kernel reports Occupancy > 1 or holds <= 256 registers (VGPRs + AGPRs), i.e. when a second wave could be
placed on its SIMD -- the configuration in which the operators gave wrong tables.

    check_p1_occupancy.py <remarks file> [--allow-occupancy N] [--expect K]
                          (N > 1: development builds with -DP1_WAVES_PER_SIMD=N; K: operator kernels the object must hold -- 7 in
                           p1_reach.o, 3 in p1_reach_tv50.o, the time-vectorised kernel alone)
"""
import re
import sys

PINNED = ("armour_p1_chain_kernel", "armour_p1_pzop_kernel", "armour_p1_tv_kernel")


def parse(text):
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: (.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            kernels[cur] = {}
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            kernels[cur][k.strip()] = v.strip()
    return kernels


def main(argv):
    allow = 1
    if "--allow-occupancy" in argv:
        allow = int(argv[argv.index("--allow-occupancy") + 1])
    expect = 7   # chain<1>, chain<3>, chain<4>, pzop, tv<1>, tv<3>, tv<4>
    if "--expect" in argv:
        expect = int(argv[argv.index("--expect") + 1])
    kernels = parse(open(argv[1]).read())
    seen, bad = 0, []
    for name, r in kernels.items():
        if not any(p in name for p in PINNED):
            continue
        seen += 1
        occ = int(r.get("Occupancy [waves/SIMD]", "0"))
        regs = int(r.get("VGPRs", "0")) + int(r.get("AGPRs", "0"))
        if occ > allow or (allow == 1 and regs <= 256):
            bad.append(f"{name}: occupancy {occ}, {r.get('VGPRs')} VGPRs + {r.get('AGPRs')} AGPRs")
    if seen < expect:
        print(f"check_p1_occupancy: only {seen} of the {expect} operator kernels found in {argv[1]}", file=sys.stderr)
        return 2
    if bad:
        print("check_p1_occupancy: these kernels would share a SIMD with a second wave:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    print(f"check_p1_occupancy: {seen} operator kernels, all at {allow} wave(s) per SIMD")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
