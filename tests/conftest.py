import ctypes
import os
import sys

import numpy as np
import torch  # noqa: F401  first, so that torch's bundled HIP runtime initialises before librfwhip.so pulls in /opt/rocm's
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

from __graft_entry__ import load_oracle, load_package  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def orc():
    o = load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def emu_lib():
    """Host-emulation build of the product sources (tests/emu/build_emu.py) — CPU tier only."""
    import build_emu
    return ctypes.CDLL(build_emu.build())


@pytest.fixture
def make_emu(pkg, emu_lib):
    def factory(rank=0, world=1):
        return pkg._binding.CoreBinding(emu_lib, "rfwhip_", 0, rank, world)
    return factory


@pytest.fixture
def make_oracle(pkg, orc):
    def factory(rank=0, world=1):
        return orc.OracleContext(pkg, rank, world)
    return factory


@pytest.fixture
def make_hip(pkg):
    """The product path: librfwhip.so on cuda:0.  Raises (never falls back) when the library or the GPU is missing."""
    def factory(rank=0, world=1):
        return pkg.RenderContext(device=0, rank=rank, world=world)
    return factory


def image_stats(a, b, tol=1e-3):
    """Per-pixel RGB L2 distance between two float RGBA images: fraction above tol, RMSE."""
    d = np.sqrt(((a[..., :3].astype(np.float64) - b[..., :3]) ** 2).sum(-1))
    return float((d > tol).mean()), float(np.sqrt((d ** 2).mean())), d
