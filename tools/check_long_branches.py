#!/usr/bin/env python3
"""Build-time check (armour_amd/csrc/Makefile): no device function of an object relaxes a branch through an UNSAVED return address.

A branch over more than 128 KB is expanded by the compiler into  s_getpc_b64 s[a:b] / s_add_u32 / s_addc_u32 / s_setpc_b64 s[a:b]  on a scavenged
scalar pair; in a function that makes no calls ROCm 7.2's clang takes s[30:31] -- the return address -- without saving it, and a function that has
taken such a branch returns into its own middle (pz_wave.h PZ_KEEP_RETURN_ADDRESS).  This script disassembles every gfx950 code
object of the given object / library files and fails when a function contains such a sequence on s[30:31] and neither saves s30 / s31 to a VGPR
lane (v_writelane_b32) nor copies the pair elsewhere first.

    check_long_branches.py file.o|file.so [...]
"""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def code_objects(path):
    """gfx950 ELF images inside a host object / shared library (offload bundles are located by their ELF magic + e_machine AMDGPU)."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(b"\x7fELF\x02\x01\x01", pos)
        if pos < 0:
            break
        if data[pos + 18:pos + 20] == b"\xe0\x00":   # EM_AMDGPU = 224
            # size from the section header table: e_shoff + e_shnum * e_shentsize
            shoff = int.from_bytes(data[pos + 40:pos + 48], "little")
            shentsize = int.from_bytes(data[pos + 58:pos + 60], "little")
            shnum = int.from_bytes(data[pos + 60:pos + 62], "little")
            out.append(data[pos:pos + shoff + shnum * shentsize])
        pos += 4
    return out


def scan(dis):
    """(functions that relax a branch through an unsaved s[30:31]: [(name, branches, lines)], functions seen, relaxed branches seen) of one
    llvm-objdump -d listing"""
    bad, nfun, nlong = [], 0, 0
    name, body = None, []

    def flush():
        nonlocal nfun, nlong
        if name is None:
            return
        nfun += 1
        lb = sum(1 for l in body if re.match(r"\s*s_getpc_b64 s\[30:31\]", l))
        if not lb:
            return
        nlong += lb
        saved = any(re.match(r"\s*(v_writelane_b32 v\d+, s3[01],|s_mov_b64 s\[\d+:\d+\], s\[30:31\]|s_mov_b32 s\d+, s3[01]\b)", l) for l in body)
        if not saved:
            bad.append((name, lb, len(body)))
    for l in dis.split("\n"):
        m = re.match(r"^[0-9a-f]{16} <(.+)>:", l)
        if m:
            flush()
            name, body = m.group(1), []
        else:
            body.append(l)
    flush()
    return bad, nfun, nlong


class NothingChecked(Exception):
    """The check could not look at any device code (a gate that passes when it has checked nothing is no gate)."""


def check(path):
    bad, nfun, nlong = [], 0, 0
    imgs = code_objects(path)
    if not imgs:
        raise NothingChecked(f"{path}: no gfx950 code object found (missing file contents, or a compressed offload bundle)")
    for k, img in enumerate(imgs):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(img)
            tmp = f.name
        try:
            r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", tmp], capture_output=True, text=True)
        except OSError as e:
            raise NothingChecked(f"{path}: cannot run {OBJDUMP}: {e}")
        finally:
            os.unlink(tmp)
        if r.returncode != 0:
            raise NothingChecked(f"{path}: {OBJDUMP} failed on code object {k} (rc {r.returncode}): {r.stderr[-200:]}")
        b, nf, nl = scan(r.stdout)
        bad += b; nfun += nf; nlong += nl
    if nfun == 0:
        raise NothingChecked(f"{path}: {len(imgs)} code object(s) disassembled to no function at all")
    return bad, nfun, nlong


def main(argv):
    rc = 0
    if len(argv) < 2:
        print(__doc__, file=sys.stderr)
        return 2
    for path in argv[1:]:
        try:
            bad, nfun, nlong = check(path)
        except (NothingChecked, OSError) as e:
            print(f"check_long_branches: NOTHING CHECKED: {e}", file=sys.stderr)
            rc = 1
            continue
        if bad:
            rc = 1
            for name, lb, n in bad:
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                print(f"check_long_branches: {path}: {d[:120]} ({n} lines) relaxes {lb} branch(es) through s[30:31] WITHOUT saving the return address", file=sys.stderr)
        else:
            print(f"check_long_branches: {os.path.basename(path)}: {nfun} functions, {nlong} relaxed branch(es) on s[30:31], every one in a function that keeps its return address elsewhere")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
