import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.lib()  # builds libv3d_oracle.so with gcc if it is missing
    return oracle


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def unit_sphere_cloud(rng, b, n):
    """ModelNet40-shaped synthetic cloud."""
    p = rng.standard_normal((b, n, 3)).astype(np.float32)
    p -= p.mean(axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=2).max(axis=1)[:, None, None]
    return np.ascontiguousarray(p, dtype=np.float32)


def lattice_cloud(side=6, scale=0.25, shuffle_seed=None):
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pts = g.astype(np.float32) * np.float32(scale)
    if shuffle_seed is not None:
        pts = pts[np.random.default_rng(shuffle_seed).permutation(len(pts))]
    return np.ascontiguousarray(pts[None])
