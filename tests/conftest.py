import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check")


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).copy()


@pytest.fixture(scope="session")
def kitti_pair():
    return (read_pgm(os.path.join(GOLDEN, "kitti00_left_1241x376.pgm")),
            read_pgm(os.path.join(GOLDEN, "kitti00_right_1241x376.pgm")))


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "oracle_golden_v1.npz"), allow_pickle=False)
