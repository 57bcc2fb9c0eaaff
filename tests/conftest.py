"""pytest configuration: `gpu` marker (tests that need an MI355X) and shared helpers.

CPU suite (`-m "not gpu"`): oracle vs known answers / golden fixtures, host logic, C-ABI symbol
export.  GPU suite (`-m gpu`): HIP path vs oracle, bit-exact, through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def ob():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def synth(pkg):
    from apd_mvs_amd import synth as s
    return s


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The product library on a GPU box; fails loudly (no fallback) when the GPU or the .so is missing."""
    assert os.path.exists(pkg.library_path()), "HIP library not built: run __graft_entry__.build()"
    assert pkg.device_count() >= 1, "no HIP device visible"
    return pkg
