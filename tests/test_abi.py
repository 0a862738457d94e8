"""The C-ABI library builds for gfx950, loads, and exports every symbol include/sniffles_amd.h declares.
No compute calls (no GPU in the CPU tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    from sniffles_amd import build
    return build.build()


def header_functions():
    src = open(os.path.join(ROOT, "include", "sniffles_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snf_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(so):
    lib = C.CDLL(so)
    fns = header_functions()
    assert len(fns) >= 14
    for f in fns:
        assert hasattr(lib, f), f"{f} declared in include/sniffles_amd.h but not exported"


def test_abi_version_and_struct_sizes(so):
    from sniffles_amd import abi, lib
    L = lib.load()
    assert L.snf_abi_version() == abi.ABI_VERSION
    # the ctypes mirrors must have the C layout (checked against sizes compiled into the oracle build)
    assert C.sizeof(abi.snf_call_t) % 8 == 0
    assert abi.CALL_DTYPE.itemsize == C.sizeof(abi.snf_call_t)


def test_no_device_fails_loudly(so):
    """Without a HIP device the product path must raise - there is no CPU fallback."""
    from sniffles_amd import lib
    from sniffles_amd.config import SnifflesConfig
    import cases
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.SnifflesAmdError, match="no HIP device"):
        lib.Batch(SnifflesConfig(), [cases.case_resplit_wrap()])


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "sniffles_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "snf_oracle" not in txt and "ref_harness" not in txt, f


def test_more_than_65535_tasks_in_one_batch_are_refused():
    """Documented limit of a batch (include/sniffles_amd.h): the task index travels in 16 bits of the packed
    keys.  The upload fails with the reason - nothing is truncated; several batches serve larger jobs."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import lib, synth
    from sniffles_amd.config import SnifflesConfig
    ti = synth.gen_fuzz(3, task_id=0)
    with pytest.raises(lib.SnifflesAmdError, match="too many tasks in one batch"):
        lib.Batch(SnifflesConfig(), [ti] * 65536)
