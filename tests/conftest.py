import os
import sys

import pytest

# The CPU oracle parallelises over time steps with OpenMP.  On a shared many-core host, "all cores" threads spinning at
# barriers next to other tenants made single tests take minutes: the tests cap it (bench.py's cpu_baseline does not).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A GPU test that stops returning (seen once in round 2: a pinned-memory launch on a freshly started box) must fail,
    # not hold the whole run until the box's watchdog ends it: five minutes per test, enforced by pytest-timeout where the
    # plugin is installed (method "thread": a call stuck inside the HIP runtime does not react to signals).
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
                item.add_marker(pytest.mark.timeout(300, method="thread"))


@pytest.fixture(scope="session")
def sample_problem():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import SAMPLE_PROBLEM
    return SAMPLE_PROBLEM
