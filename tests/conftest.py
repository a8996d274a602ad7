"""Shared fixtures.  `-m gpu` tests need a real MI355X; everything else runs on CPU."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

rt = importlib.import_module("raytrace-miniapp_amd")
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / nb) if nb > 0 else float(np.linalg.norm(a))


@pytest.fixture(scope="session")
def ase_small():
    return rt.datfile.load(GOLDEN / "ASE_small.dat.xz")


@pytest.fixture(scope="session")
def seed_small():
    return rt.datfile.load(GOLDEN / "seed_small.dat.xz")


@pytest.fixture(scope="session")
def ase_ref():
    return np.load(GOLDEN / "ASE_small_ref_cpu.npz")


@pytest.fixture(scope="session")
def seed_ref():
    return np.load(GOLDEN / "seed_small_ref_cpu.npz")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle, build
    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly if it is not built."""
    from importlib import import_module
    # Tests that hand torch device buffers to the plan need ONE HIP runtime in the process: torch brings
    # its own copy of libamdhip64, and whichever is loaded first serves both (bench.py loads torch first
    # for the same reason).  So torch goes first here too.
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    backend = import_module("raytrace-miniapp_amd.backend")
    return backend
