"""The code-generation hazards of rounds 1-4 and what holds them (profiles/r05_codegen_hazards.txt).

Round 5 found the mechanism of the memory-aperture violations (the controller's non-inlined form of round 3, the time-vectorised kernel's one-wave
blocks on narrow rows): ROCm 7.2's clang relaxes branches over more than 128 KB through s[30:31] -- the return address -- in functions that make no
calls, without saving it.  Every non-inlined device function now keeps its return address elsewhere (pz_wave.h PZ_KEEP_RETURN_ADDRESS) and `make`
runs tools/check_long_branches.py on every library it links.  Two hazards are still unexplained; what holds them is under test here:

  * reach-set operators (round 1: a memory fault under interprocedural register allocation; round 2: wrong tables with two waves per SIMD):
    the shipped objects are pinned to one wave per SIMD, `make` fails otherwise (tools/check_p1_occupancy.py); they were built with
    -enable-ipra=false until the end of round 6 -- that fence is gone (profiles/r06_ipra.txt: the compiler collects a callee's register usage AFTER
    branch relaxation, and today's source passes everything below with IPRA on),
    and HERE the same source with every LDS / arena index of the product merge and of the reduce passes range-checked (-DDBG_BOUNDS,
    armour_amd/lib/libarmour_hip_checked.so) runs a fuzz set in every launch shape: no check fires (a firing check fails the build call with
    flag 128) and the tables equal the shipped library's bit for bit; the two-waves-per-SIMD build of today's source (`make occ2`) must
    reproduce the shipped tables over 12 fresh handles per batch size;
  * controller kernels: `make` fails unless controller.o holds exactly the two kernels, each with a static stack (tools/check_controller_codegen.py);
    HERE the check is run on the remarks of the object that was shipped, and on a doctored copy that it must reject.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "armour_amd", "lib", "libarmour_hip_checked.so")

_FUZZ = r'''
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
print("library", _lib.LIB_PATH, flush=True)
cases = [(1, 3, {}), (1, 3, {_lib.OPT_P1_STEP_WAVES: 1}), (1, 3, {_lib.OPT_P1_STEP_WAVES: 3}), (2, 2, {_lib.OPT_P1_STEP_FREE: 0}),
         (6, 2, {_lib.OPT_P1_BUILD: 1}), (20, 1, {_lib.OPT_P1_BUILD: 1, _lib.OPT_P1_STEP_WAVES: 1}),
         (6, 2, {_lib.OPT_P1_BUILD: 2}), (20, 1, {_lib.OPT_P1_BUILD: 2, _lib.OPT_P1_TV_WAVES: 3}), (24, 1, {_lib.OPT_P1_BUILD: 2, _lib.OPT_P1_TV_WAVES: 1})]
for seed in range(%d):
    for B, O, opts in cases:
        bp = random_batch(3000 + 17 * seed + B, B, O)
        if seed %% 2:   # fast starts: the largest products
            bp["qd0"][B - 1] = 0.9 * np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])
        nlp = ArmourNLP(T=100)
        for o, v in opts.items():
            nlp.set_option(o, v)
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])   # raises if a range check fired (flag 128)
        h = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes())
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in (0, 41, 99):
                    for a in nlp.pz(which, i, t, b=B - 1):
                        h.update(np.ascontiguousarray(a).tobytes())
        info = nlp.build_info()
        print("case", seed, B, sorted(opts.items()), info["kernel"], info["waves"], h.hexdigest(), flush=True)
        nlp.close()
print("fuzz done", flush=True)
'''


def _run(lib, seeds):
    env = dict(os.environ)
    if lib:
        env["ARMOUR_HIP_LIB"] = lib          # (the library path is the one thing the environment still selects)
    else:
        env.pop("ARMOUR_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", _FUZZ % (ROOT, seeds)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz done" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return [line for line in r.stdout.splitlines() if line.startswith("case ")], r.stdout.splitlines()[0]


@pytest.mark.gpu
def test_range_checked_operators_run_the_fuzz_set_clean_and_agree_with_the_shipped_library():
    assert os.path.exists(CHECKED), "build it: make -C armour_amd/csrc checked (part of `make all`)"
    checked, lib_line = _run(CHECKED, 4)
    assert "libarmour_hip_checked.so" in lib_line
    shipped, lib_line2 = _run(None, 4)
    assert lib_line2.endswith("libarmour_hip.so")
    assert len(checked) == 36 and checked == shipped


@pytest.mark.gpu
def test_two_waves_per_simd_build_reproduces_the_shipped_tables():
    """The round-2 hazard (wrong tables every second launch with two waves per SIMD) as a regression test: `make -C armour_amd/csrc occ2`
    builds the shipped source at occupancy 2; the per-step kernel is forced, 12 fresh handles per batch size build the same worlds, and
    every digest (tables, g, Jacobian) must equal the shipped library's (tools/ab.py; profiles/r05_codegen_hazards.txt)."""
    occ2 = os.path.join(ROOT, "armour_amd", "lib", "libarmour_hip_occ2.so")
    if not os.path.exists(occ2):
        pytest.skip("libarmour_hip_occ2.so not built (make -C armour_amd/csrc occ2)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab.py"), "tree@1=1", "occ2@1=1", "--rounds", "1", "--reps", "12", "--", "16", "64", "128"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "digests identical across variants, rounds and repeats: True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_checked_library_exports_the_whole_abi():
    """CPU: the range-checked library is a full libarmour_hip (same objects but p1_reach.o) -- every symbol of include/armour_hip.h."""
    import ctypes as C
    from armour_amd import _lib
    if not os.path.exists(CHECKED):
        pytest.skip("libarmour_hip_checked.so not built (make -C armour_amd/csrc)")
    L = C.CDLL(CHECKED)
    for name in _lib.EXPORTS:
        assert hasattr(L, name), name


def test_controller_codegen_check_accepts_the_shipped_object_and_rejects_a_callee(tmp_path):
    """CPU: tools/check_controller_codegen.py on the remarks `make` kept of the shipped controller.o (exactly two kernels, static stacks), and
    on a copy with a third function / a dynamic stack, which it must refuse."""
    remarks = os.path.join(ROOT, "armour_amd", "lib", "controller.resources.txt")
    if not os.path.exists(remarks):
        pytest.skip("no build remarks (make -C armour_amd/csrc)")
    tool = os.path.join(ROOT, "tools", "check_controller_codegen.py")
    ok = subprocess.run([sys.executable, tool, remarks], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    text = open(remarks).read()
    extra = tmp_path / "extra.txt"
    extra.write_text(text + "\ncontroller_core.h:1:1: remark: Function Name: _ZN3ctl13interval_rneaEv [-Rpass-analysis=kernel-resource-usage]\n"
                            "controller_core.h:1:1: remark:     ScratchSize [bytes/lane]: 512 [-Rpass-analysis=kernel-resource-usage]\n")
    assert subprocess.run([sys.executable, tool, str(extra)], capture_output=True, text=True).returncode == 1
    dyn = tmp_path / "dyn.txt"
    dyn.write_text(text.replace("Dynamic Stack: False", "Dynamic Stack: True", 1))
    assert subprocess.run([sys.executable, tool, str(dyn)], capture_output=True, text=True).returncode == 1


def test_long_branch_check_on_listings_and_on_the_shipped_library():
    """CPU: tools/check_long_branches.py, the build gate of DESIGN.md 7.  On listings: a leaf function that relaxes a branch through s[30:31]
    without saving it is reported; the same function with the return address kept in a VGPR lane, a function that only CALLS through s[30:31]'s
    neighbours, and one without relaxed branches are not.  On the shipped library: nothing to report."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_long_branches as clb
    leaf = """
0000000000001000 <leaf_without_a_saved_return_address>:
\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)
\ts_cbranch_scc0 8
\ts_getpc_b64 s[30:31]
\ts_add_u32 s30, s30, 0x34f88
\ts_addc_u32 s31, s31, 0
\ts_setpc_b64 s[30:31]
\tv_mov_b32_e32 v0, 0
\ts_setpc_b64 s[30:31]
"""
    kept = leaf.replace("leaf_without_a_saved_return_address", "leaf_with_the_guard").replace("\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)\n",
                                                                                              "\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)\n\tv_writelane_b32 v40, s30, 0\n\tv_writelane_b32 v40, s31, 1\n")
    other = """
0000000000003000 <calls_through_another_pair>:
\ts_getpc_b64 s[16:17]
\ts_add_u32 s16, s16, 0x100
\ts_addc_u32 s17, s17, 0
\ts_swappc_b64 s[30:31], s[16:17]
\ts_setpc_b64 s[30:31]
"""
    bad, nfun, nlong = clb.scan(leaf + kept + other)
    assert nfun == 3 and nlong == 2 and [b[0] for b in bad] == ["leaf_without_a_saved_return_address"], (bad, nfun, nlong)
    lib = os.path.join(ROOT, "armour_amd", "lib", "libarmour_hip.so")
    if os.path.exists(lib) and os.path.exists(clb.OBJDUMP):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_long_branches.py"), lib], capture_output=True, text=True)
        assert r.returncode == 0 and "0 relaxed branch(es)" in r.stdout, r.stdout + r.stderr
        nfun = int(re.search(r"(\d+) functions", r.stdout).group(1))
        assert nfun >= 100, r.stdout        # ("0 functions, 0 relaxed branches" must not read as a pass)
    # a file without device code, and a missing disassembler, are failures of the gate -- not passes
    empty = os.path.join(ROOT, "tools", "check_long_branches.py")
    r = subprocess.run([sys.executable, empty, empty], capture_output=True, text=True)
    assert r.returncode != 0 and "NOTHING CHECKED" in r.stderr, r.stdout + r.stderr
    if os.path.exists(lib):
        saved = clb.OBJDUMP
        try:
            clb.OBJDUMP = "/nonexistent/llvm-objdump"
            with pytest.raises(clb.NothingChecked):
                clb.check(lib)
        finally:
            clb.OBJDUMP = saved
