"""pytest plumbing: markers, import paths, library fixtures.

  -m "not gpu": oracle vs golden vectors / vs the live third-party reference, host logic, the C-ABI
                symbol check, and the kernel + engine SOURCES executed on the CPU SIMT emulator
                (tests/simt_emu: test infrastructure, never a product path).
  -m gpu:       the parity tests proper: libneutts_hip.so on a real MI355X, through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neutts-air_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes on CPU; deselected by default via -m")


def _load_build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ntts_build", os.path.join(PKG, "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def emu_lib():
    """libneutts_emu.so: the product sources compiled against the SIMT emulator headers."""
    return _load_build().build_emu(verbose=False)


@pytest.fixture(scope="session")
def hip_lib():
    """libneutts_hip.so (gfx950).  Built here if missing; never falls back to anything else."""
    return _load_build().build(verbose=False)
