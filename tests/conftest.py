import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_selection(request):
    """Every test starts on the real library; the host-tier tests switch to theirs by asking for it (`emu.emu.lib()` calls
    `sniffles_amd.lib.use_library`)."""
    from sniffles_amd import lib
    lib.use_library(None)
    yield
    lib.use_library(None)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # test infrastructure (oracle/oracle.py)
    oracle.build()
    return oracle


def pytest_collection_modifyitems(config, items):
    """GPU runs: bring up PyTorch's HIP runtime before the library's.  torch ships its own copy of the runtime; when the
    library's copy (from /opt/rocm) has claimed the device first, torch's later `torch.cuda` initialisation finds no GPU
    (seen when tests/test_output_modes.py ran on its own) - the order torch-first is the one every product entry point has
    (bench.py, `__graft_entry__`)."""
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        return
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # noqa: BLE001 - no torch / no GPU: the GPU tests themselves say so
            pass
