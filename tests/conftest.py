import os
import sys

import numpy as np
import pytest

try:                        # before anything loads libnbody_amd.so: a process that uses both must bring torch's bundled HIP
    import torch  # noqa: F401   runtime up first (n-body_amd/__init__.py: lib()); some tests here hand torch streams and
except ImportError:         # tensors to the library, and which test file happens to be collected first must not matter
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Built artefacts are git-ignored.  In a working tree they are there (the driver's build() ran); in a fresh
    # checkout build them once (hipcc cross-compiles without a GPU) instead of failing every test on a missing file.
    needed = [os.path.join(ROOT, "n-body_amd", "lib", "libnbody_amd.so"), os.path.join(ROOT, "n-body_amd", "bin", "nbody_benchmark"),
              os.path.join(ROOT, "n-body_amd", "bin", "nbody_sim"), os.path.join(ROOT, "oracle", "libnbody_oracle.so")]
    if not all(os.path.exists(f) for f in needed):
        entry.build()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    return entry.load_oracle()


@pytest.fixture(scope="session")
def nb():
    """The product package (ctypes over the C-ABI)."""
    return entry.load_package()


def load_golden(O, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["n"])
    b = O.Bodies(n)
    b.m[:] = g["m"][0]
    b.radius[:] = g["radius"][0]
    for k in ("posX", "posY", "velX", "velY"):
        getattr(b, k)[:] = g["init_" + k]
    return g, b


def rel_err(ax, ay, rx, ry):
    """per-body ||a - ref|| / ||ref||"""
    den = np.hypot(rx, ry)
    return np.hypot(ax - rx, ay - ry) / np.where(den > 0, den, 1.0)


@pytest.fixture(scope="session")
def golden():
    return load_golden
