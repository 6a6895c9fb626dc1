"""The round-3 boundary entries (include/armour_hip.h): reduced outputs (armour_eval_violations*), the in-process
multi-device batch (armour_batch_*) and the per-handle build option.

CPU part: the partition arithmetic equals armour_amd/sharding.py's (the one bench.py shards ranks with).
GPU part, all through the C ABI:
  * the violation record of every problem equals what numpy computes from the full g and the bounds
    (RT/NLPclass.cu:422-538 restated in armour_check_feasible), including the verdict, for feasible and infeasible points;
  * a batch over device slots [0, 0] (two handles, two host threads, one device -- what a 1-GPU box can exercise) gives
    bit-identical g / jac / violation records / solver results to ONE handle holding the same problems, and its bounds
    gather in place;
  * ARMOUR_OPT_P1_BUILD = 1 and 2 reproduce each other's keys / coefficients bit for bit and the radii to the stated
    1e-12 (the tolerance contract of the header), and option 1 makes a world's tables independent of the batch size."""
import numpy as np
import pytest


def test_partition_matches_sharding_rule():
    from armour_amd.planner import batch_partition
    from armour_amd.sharding import shard_range
    for B in (1, 2, 7, 8, 9, 127, 128, 1000, 1024):
        for G in (1, 2, 3, 4, 8):
            first = batch_partition(B, G)
            assert first[0] == 0 and first[-1] == B and len(first) == G + 1
            for r in range(G):
                assert (first[r], first[r + 1]) == shard_range(B, r, G)
            sizes = np.diff(first)
            assert sizes.max() - sizes.min() <= 1 and (sizes >= 0).all()


def test_partition_rejects_bad_arguments():
    import ctypes as C
    from armour_amd import _lib
    L = _lib.load()
    first = (C.c_int32 * 4)()
    assert L.armour_batch_partition(5, 0, first) == _lib.EINVAL
    assert L.armour_batch_partition(-1, 2, first) == _lib.EINVAL


def _reference_records(nlp, g):
    """numpy restatement of the record from the full g (the checker)."""
    _, _, gl, gu = nlp.get_bounds_info()
    feas = nlp.finalize_solution(g)
    out = []
    for b in range(nlp.B):
        viol = np.maximum(0.0, np.maximum(gl[b] - g[b], g[b] - gu[b]))
        wr = int(np.argmax(viol)) if viol.max() > 0 else -1
        out.append(dict(l1=float(viol.sum()), worst=float(viol.max()), worst_row=wr, n_violated=int((viol > 0).sum()), feasible=bool(feas[b])))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("B,O", [(1, 20), (5, 7), (40, 12)])
def test_violation_records_equal_the_row_test_on_the_full_g(B, O):
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T = 100 if B < 40 else 60
    bp = random_batch(4200 + B, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    for s, scale in enumerate((0.0, 0.3, 1.0)):   # x = 0 (often feasible), small and full-range points
        k = random_k(900 + s, B) * scale
        g, _ = nlp.eval_g_jac(k)
        ref = _reference_records(nlp, g)
        rec = nlp.eval_violations(k)
        for b in range(B):
            r, e = rec[b], ref[b]
            assert r["feasible"] == e["feasible"] and (r["n_outside_slack"] == 0) == e["feasible"], (b, r, e)
            assert r["n_violated"] == e["n_violated"] and r["worst_row"] == e["worst_row"], (b, r, e)
            assert r["worst"] == e["worst"]                                  # one row's value: bit-equal
            assert abs(r["l1_violation"] - e["l1"]) <= 1e-12 * max(1.0, e["l1"])   # a sum in another order
        again = nlp.eval_violations(k)
        assert again == rec                                                  # fixed summation order: bit-reproducible
    nlp.close()


@pytest.mark.gpu
def test_violation_records_through_the_device_entry():
    import ctypes as C
    import torch
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    B, O, T = 6, 9, 100
    bp = random_batch(77, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    k = random_k(5, B)
    d_k = torch.tensor(k, device="cuda")
    d_out = torch.zeros(B * C.sizeof(_lib.ArmourViolation), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    nlp.eval_violations_device(d_k.data_ptr(), d_out.data_ptr(), st.cuda_stream)
    st.synchronize()
    raw = d_out.cpu().numpy().tobytes()
    recs = (_lib.ArmourViolation * B).from_buffer_copy(raw)
    host = nlp.eval_violations(k)
    for b in range(B):
        assert (recs[b].l1_violation, recs[b].worst, recs[b].worst_row, recs[b].n_violated, recs[b].n_outside_slack, bool(recs[b].feasible)) == \
               (host[b]["l1_violation"], host[b]["worst"], host[b]["worst_row"], host[b]["n_violated"], host[b]["n_outside_slack"], host[b]["feasible"])
    nlp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [[0], [0, 0], [0, 0, 0]])
def test_batch_over_device_slots_equals_one_handle(slots):
    from armour_amd import _lib
    from armour_amd.planner import ArmourBatchNLP, ArmourNLP, batch_partition
    from armour_amd.worlds import random_batch, random_k
    B, O, T = 7, 8, 100
    bp = random_batch(31, B, O)
    one = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    bt = ArmourBatchNLP(slots, T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    assert (bt.B, bt.m, bt.n) == (one.B, one.m, one.n)
    assert batch_partition(B, len(slots))[-1] == B
    k = random_k(3, B)
    g1, j1 = one.eval_g_jac(k)
    g2, j2 = bt.eval_g_jac(k)
    assert np.array_equal(g1, g2) and np.array_equal(j1, j2)          # same kernels on the same per-problem tables
    b1, b2 = one.get_bounds_info(), bt.get_bounds_info()
    for a, c in zip(b1, b2):
        assert np.array_equal(a, c)
    assert one.eval_violations(k) == bt.eval_violations(k)
    s1, s2 = one.solve(), bt.solve()
    for r1, r2 in zip(s1, s2):
        assert np.array_equal(r1["k_opt"], r2["k_opt"]) and r1["feasible"] == r2["feasible"] and r1["status"] == r2["status"]
        assert r1["cost"] == r2["cost"] and r1["iterations"] == r2["iterations"]
    assert bt.build_ms > 0
    assert np.array_equal(bt.prune_margin(), one.prune_margin())     # every problem's prune margin, whichever slot built it (both built step by step here)
    # the row lists are per handle: culled row test and culled solve on every slot (round 5), same records and results
    bt.set_option(_lib.OPT_CULL_ROWS, 1); bt.set_option(_lib.OPT_SOLVE_CULL, 1); bt.set_option(_lib.OPT_SOLVE_DEVICE, 2)
    assert [(v["l1_violation"], v["n_violated"], v["feasible"]) for v in one.eval_violations(k)] == [(v["l1_violation"], v["n_violated"], v["feasible"]) for v in bt.eval_violations(k)]
    for r1, r2 in zip(s1, bt.solve()):
        assert np.array_equal(r1["k_opt"], r2["k_opt"]) and r1["feasible"] == r2["feasible"] and r1["status"] == r2["status"] and r1["cost"] == r2["cost"]
    # errors of a worker thread reach the caller: an out-of-range capacity on one slot's problems
    with pytest.raises(_lib.ArmourError):
        bad = bp["q0"].copy(); bad[B - 1, 0] = np.nan
        bt.set_parameters(bad, bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    one.close(); bt.close()


@pytest.mark.gpu
def test_build_option_pins_the_kernel_and_states_the_tolerance():
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    B, O, T = 3, 4, 100
    bp = random_batch(611, B, O)
    per_step = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    tv = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 2).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    single = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"][1], bp["qd0"][1], bp["qdd0"][1], bp["q_des"][1], bp["obstacles"][1])
    for which, cnt in (("link", per_step.J), ("torque", per_step.n)):
        for i in range(cnt):
            for t in range(0, T, 7):
                for b in range(B):
                    c1, r1, k1, co1 = per_step.pz(which, i, t, b=b)
                    c2, r2, k2, co2 = tv.pz(which, i, t, b=b)
                    assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2)   # bit for bit
                    assert np.abs(r1 - r2).max() <= 1e-12                                                   # the contract
                cs, rs, ks, cos = single.pz(which, i, t)
                c1, r1, k1, co1 = per_step.pz(which, i, t, b=1)
                assert np.array_equal(ks, k1) and np.array_equal(cos, co1) and np.array_equal(cs, c1) and np.array_equal(rs, r1)
    assert np.abs(per_step.torque_radius() - tv.torque_radius()).max() <= 1e-12
    assert np.array_equal(per_step.torque_radius()[1], single.torque_radius()[0])   # option 1: independent of the batch size
    with pytest.raises(_lib.ArmourError):
        per_step.set_option(_lib.OPT_P1_BUILD, 3)
    per_step.close(); tv.close(); single.close()


@pytest.mark.gpu
def test_work_memory_cap_changes_the_block_count_not_the_tables():
    """ARMOUR_OPT_P1_WORK_MEMORY_MB: the time-vectorised build on fewer blocks (they loop over the groups) gives the same tables bit for
    bit; a cap below one block's slots sends the batch to the step-by-step kernel (radii to the tolerance contract)."""
    import hashlib
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    B, O, T = 24, 3, 100
    bp = random_batch(77, B, O)
    ks = random_k(5, B)

    def build(cap_mb):
        nlp = ArmourNLP(T=T)
        if cap_mb is not None:
            nlp.set_option(_lib.OPT_P1_WORK_MEMORY_MB, cap_mb)
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        g, jac = nlp.eval_g_jac(ks)
        digest = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()
                              + np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(jac).tobytes()).hexdigest()
        info, tr = nlp.build_info(), nlp.torque_radius().copy()
        nlp.close()
        return digest, info, tr

    free_digest, free_info, free_tr = build(None)
    assert free_info["kernel"] == "time_vectorised" and free_info["launches"] == 1
    capped_digest, capped_info, _ = build(5 * 120)            # room for five blocks of ~112 MiB: 48 groups in ten rounds
    assert capped_info["kernel"] == "time_vectorised" and capped_digest == free_digest
    tiny_digest, tiny_info, tiny_tr = build(50)               # not one block: step by step
    assert tiny_info["kernel"] == "per_step" and np.abs(tiny_tr - free_tr).max() <= 1e-12
    with pytest.raises(_lib.ArmourError):
        ArmourNLP(T=T).set_option(_lib.OPT_P1_WORK_MEMORY_MB, -1)


@pytest.mark.gpu
def test_all_slots_of_a_problem_set_are_built_by_one_kernel():
    """31 problems over two slots are 16 + 15, and with every slot choosing for itself the first would be built time-vectorised
    (16 * 100 >= 50 * 31, the threshold since the end of round 4) and the second step by step -- radii different to 1e-12 WITHIN one batch.  Now the batch takes the kernel its
    smallest shard would take for all slots; pinned by the caller, the pinned kernel.  The step-by-step tables are those of one handle
    held to that kernel, bit for bit (they do not depend on the batch mates)."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourBatchNLP, ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    B, O, T = 31, 2, 100
    bp = random_batch(1234, B, O)
    k = random_k(8, B)
    bt = ArmourBatchNLP([0, 0], T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    kinds = {i["kernel"] for i in bt.build_info() if i}
    assert kinds == {"per_step"}, bt.build_info()
    one = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    g1, j1 = one.eval_g_jac(k)
    g2, j2 = bt.eval_g_jac(k)
    assert np.array_equal(g1, g2) and np.array_equal(j1, j2)
    auto = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])   # one handle of 31: time-vectorised
    assert auto.build_info()["kernel"] == "time_vectorised"
    g3, j3 = auto.eval_g_jac(k)
    assert np.abs(g3 - g2).max() <= 1e-12 * max(1.0, np.abs(g2).max()) and np.abs(j3 - j2).max() <= 1e-12 * max(1.0, np.abs(j2).max())
    bt.set_option(_lib.OPT_P1_BUILD, 2).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    assert {i["kernel"] for i in bt.build_info() if i} == {"time_vectorised"}
    bt.set_option(_lib.OPT_P1_BUILD, 0).set_parameters(bp["q0"][:4], bp["qd0"][:4], bp["qdd0"][:4], bp["q_des"][:4], bp["obstacles"][:4])
    assert {i["kernel"] for i in bt.build_info() if i} == {"per_step"}
    one.close(); auto.close(); bt.close()


@pytest.mark.gpu
def test_eight_slots_of_one_device_host_overhead_per_slot():
    """The only multi-device evidence a 1-GPU box allows: eight device slots of device 0 -- eight handles, eight
    streams, eight persistent worker threads -- behind ONE caller thread.  Results equal one handle's bit for bit, and what the batch
    layer adds on the host per call and slot (condition-variable hand-off to the slot's worker and back) stays below 60 us: the reduced-
    output evaluation of one small problem per slot, median of 200 calls, against the same call on a single handle."""
    import time
    from armour_amd.planner import ArmourBatchNLP, ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    S, O, T = 8, 2, 20
    bp = random_batch(55, S, O)
    k = random_k(2, S)
    bt = ArmourBatchNLP([0] * S, T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    one = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    lone = ArmourNLP(T=T).set_parameters(bp["q0"][0], bp["qd0"][0], bp["qdd0"][0], bp["q_des"][0], bp["obstacles"][0])
    assert bt.eval_violations(k) == one.eval_violations(k)
    ga, ja = bt.eval_g_jac(k)
    gb, jb = one.eval_g_jac(k)
    assert np.array_equal(ga, gb) and np.array_equal(ja, jb)

    def median_us(fn, reps=200):
        for _ in range(20):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e6
    t_batch = median_us(lambda: bt.eval_violations(k))
    t_lone = median_us(lambda: lone.eval_violations(k[:1]))
    per_slot = (t_batch - t_lone) / S
    print(f"eight slots of device 0: {t_batch:.1f} us per batch call, one handle with one problem {t_lone:.1f} us -> {per_slot:.1f} us per slot on top")
    assert per_slot < 60.0, (t_batch, t_lone)
    bt.close(); one.close(); lone.close()


@pytest.mark.gpu
def test_two_handles_building_large_batches_at_once_share_the_work_memory():
    """the time-vectorised build's work slots (111.6 MiB per block, 28.6 GiB for 128 problems) are ONE arena per
    device shared by the handles of the process, held by a build only while it runs.  Two handles building 128 problems each from two
    host threads at the same time: identical tables to a build alone, and the device never holds more than 35 GiB beyond what it held
    before (sampled every 2 ms from a third thread) -- round 3 needed 2 x 28.6.  ARMOUR_OPT_P1_KEEP_WORK_MEMORY = 0 releases the arena
    after every build of that handle; by default it goes with the device's last handle."""
    import hashlib, threading, time
    import ctypes as C
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    L = _lib.load()
    def used_bytes():
        free, total = C.c_uint64(), C.c_uint64()
        _lib.check(L.armour_device_memory(0, C.byref(free), C.byref(total)))
        return total.value - free.value
    gib = lambda x: x / 2.0**30
    B, O, T = 128, 2, 100
    batches = [random_batch(9000 + i, B, O) for i in range(2)]
    def digest(nlp):
        return hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()).hexdigest()
    start = used_bytes()
    alone = []
    for bp in batches:
        nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        assert nlp.build_info()["kernel"] == "time_vectorised"
        alone.append(digest(nlp)); nlp.close()
    assert gib(used_bytes() - start) <= 1.0            # the last handle of the device took the arena with it
    handles = [ArmourNLP(T=T).set_option(_lib.OPT_P1_KEEP_WORK_MEMORY, 0) for _ in batches]
    for h, bp in zip(handles, batches):                 # first builds: tables allocated, code objects loaded
        h.set_parameters(bp["q0"][:2], bp["qd0"][:2], bp["qdd0"][:2], bp["q_des"][:2], bp["obstacles"][:2])
    base = used_bytes()
    peak, stop = [base], threading.Event()
    def sample():
        while not stop.is_set():
            peak[0] = max(peak[0], used_bytes()); time.sleep(0.002)
    errors = []
    def build(h, bp):
        try:
            for _ in range(2):
                h.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    sampler = threading.Thread(target=sample); sampler.start()
    threads = [threading.Thread(target=build, args=(h, bp)) for h, bp in zip(handles, batches)]
    for t in threads: t.start()
    for t in threads: t.join()
    stop.set(); sampler.join()
    assert not errors, errors
    assert [digest(h) for h in handles] == alone
    extra_gib, after_gib = gib(peak[0] - base), gib(used_bytes() - base)
    print(f"two concurrent builds of 128 problems: peak {extra_gib:.1f} GiB above the idle handles, {after_gib:.1f} GiB still held afterwards (option 0)")
    assert extra_gib <= 35.0 and after_gib <= 4.0, (extra_gib, after_gib)
    # ... the default keeps the shared arena for the next build of ANY handle of the device, once
    for h in handles:
        h.set_option(_lib.OPT_P1_KEEP_WORK_MEMORY, 1)
    for h, bp in zip(handles, batches):
        h.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    kept = gib(used_bytes() - base)
    assert 20.0 <= kept <= 35.0, kept
    assert [digest(h) for h in handles] == alone
    handles[0].close()
    assert gib(used_bytes() - base) >= 20.0            # the other handle is still there
    handles[1].close()
    assert gib(used_bytes() - start) <= 1.0
