import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sample_problem():
    """The reference's commented sample input (RT/armour_main.cu:18-33): a known input without a known answer."""
    import numpy as np
    q0 = np.array([0.6543, -0.0876, -0.4837, -1.2278, -1.5735, -1.0720, 0])
    z = np.zeros(7)
    q_des = np.array([0.6831, 0.009488, -0.2471, -0.9777, -1.414, -0.9958, 0])
    five = np.array([
        [-0.28239, -0.33281, 0.88069, 0.069825, 0, 0, 0, 0.09508, 0, 0, 0, 0.016624],
        [-0.19033, 0.035391, 1.3032, 0.11024, 0, 0, 0, 0.025188, 0, 0, 0, 0.014342],
        [0.67593, -0.085841, 0.43572, 0.17408, 0, 0, 0, 0.07951, 0, 0, 0, 0.18012],
        [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981],
        [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981]])
    obs = np.vstack([five[[0, 1, 2, 3, 4]], five[[0, 1, 2, 3, 4]]])
    return dict(q0=q0, qd0=z, qdd0=z, q_des=q_des, obstacles=obs)
