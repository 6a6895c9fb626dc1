#!/usr/bin/env python3
"""Build-time check (armour_amd/csrc/Makefile): the tracking controller's two kernels must compile to ONE function each, with a
static stack.  Round 3 wrote the interval RNEA's halves as functions of their own: the callees reached the kernel's frame through
generic pointers and both kernels ended in a memory-aperture violation on the MI355X (controller_core.h); the cure was
to inline everything into the kernels, and nothing but this check keeps an edit -- or a compiler that stops inlining a grown
function -- from bringing the faulting form back silently.  Reads the -Rpass-analysis=kernel-resource-usage remarks of controller.o
and fails the build when
  * the object holds a device function besides the two kernels (something was not inlined),
  * a kernel reports a dynamic stack (an indirect or recursive call survived), or
  * a kernel's scratch frame exceeds the bound below (a callee's frame the kernel would address through generic pointers).

    check_controller_codegen.py <remarks file>
"""
import re
import sys

KERNELS = ("armour_controller_kernel", "armour_controller_split_kernel")
MAX_SCRATCH_BYTES = 16384   # per lane; the shipped kernels: see the build log line this script prints


def parse(text):
    fns, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: (.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            fns[cur] = {}
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            fns[cur][k.strip()] = v.strip()
    return fns


def main(argv):
    fns = parse(open(argv[1]).read())
    bad, seen = [], []
    for name, r in fns.items():
        kernel = next((k for k in KERNELS if k in name and (k != "armour_controller_kernel" or "split" not in name)), None)
        if kernel is None:
            bad.append(f"{name}: a device function of its own (the controller's halves must be inlined into the kernels)")
            continue
        seen.append(kernel)
        if r.get("Dynamic Stack", "False") != "False":
            bad.append(f"{name}: dynamic stack")
        scratch = int(r.get("ScratchSize [bytes/lane]", "0"))
        if scratch > MAX_SCRATCH_BYTES:
            bad.append(f"{name}: {scratch} B of scratch per lane (> {MAX_SCRATCH_BYTES})")
    if sorted(set(seen)) != sorted(KERNELS):
        print(f"check_controller_codegen: expected the kernels {KERNELS}, found {seen} in {argv[1]}", file=sys.stderr)
        return 2
    if bad:
        print("check_controller_codegen:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    print("check_controller_codegen: " + "; ".join(f"{n.split('(')[0][-40:]}: {r.get('VGPRs')} VGPRs, {r.get('ScratchSize [bytes/lane]')} B scratch, dynamic stack {r.get('Dynamic Stack')}" for n, r in fns.items()))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
