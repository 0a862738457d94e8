"""Builds the HIP library in-tree: sniffles_amd/libsniffles_amd.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the reference's double
arithmetic (CPython) has no fused multiply-add and the parity bar is bit-exactness.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libsniffles_amd.so")
SOURCES = ["snf_lib.hip", "snf_myers.hip", "snf_combine.hip", "snf_extract.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-result", "-Wno-unused-value"]


def fast_so() -> str:
    import sysconfig
    return os.path.join(HERE, "_snf_fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_fast(force: bool = False) -> str:
    """The CPython extension that turns record tables into SVCall objects (csrc/snf_pyfast.c; host-side formatting only)."""
    import sysconfig
    so, src = fast_so(), os.path.join(CSRC, "snf_pyfast.c")
    hdr = os.path.join(HERE, "..", "include", "sniffles_amd.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I", sysconfig.get_paths()["include"], src, "-o", so]
        subprocess.run(cmd, check=True)
    return so


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sniffles_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    build_fast(force)
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
