"""Real ranks on the GPU box.  No round had more than one MI355X, so this is the only multi-rank evidence obtainable: `bench.py --gpus 2` as
the driver starts it -- the parent launches one fresh process per rank BEFORE anything touches the GPU (bench.py launch_ranks) -- with both
ranks on device 0 and gloo for the barrier and the MAX reduction (RCCL refuses two ranks on one GPU; the path has no data-path collective:
DESIGN.md 6).  Every rank builds its own worlds, times its own K steps between the barriers, and rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--repeats", "5", "--headline-only",
                          "--no-cpu-baseline", "--no-sync-probe", *flags], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_of_bench_py_share_device_0():
    one = _bench("--gpus", "1", "--batch", "3")
    two = _bench("--gpus", "2", "--batch", "3", "--rank-devices", "0,0", "--dist-backend", "gloo")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["problems_per_rank"] == [3, 3] and one["config"]["problems_per_rank"] == [3]
    # value = the problems ALL ranks evaluated x steps / the slowest rank's interval
    for js, total in ((one, 3), (two, 6)):
        assert abs(js["value"] * js["ms_per_step"] * 1e-3 - total) <= 1e-6 * total, js
    # two PROCESSES on one device take turns on it (measured: 220 k against 384 k iters/s for a lone rank at this size): nothing can be asserted about
    # the aggregate beyond its being of the same order -- the scaling curve needs one device per rank, which no round has had
    assert two["value"] > 0.2 * one["value"], (one["value"], two["value"])
    assert two["roofline"]["launch_us"] > 0 and two["p1_set_problems_ms"]["device"] > 0
