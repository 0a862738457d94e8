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
         "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("SNF_EXTRA_FLAGS", "").split()   # e.g. -DSNF_CONS_PROFILE


def fast_so() -> str:
    import sysconfig
    return os.path.join(HERE, "_snf_fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _digest(paths, extra="") -> str:
    """Content hash of the sources a binary was built from (file times do not survive a snapshot copy; the hash file travels
    with the binary, so a prebuilt library is reused exactly when its sources are the ones in the tree)."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(binary: str, digest: str) -> bool:
    try:
        with open(binary + ".srchash") as f:
            return not os.path.exists(binary) or f.read().strip() != digest
    except OSError:
        return True


def _stamp(binary: str, digest: str) -> None:
    with open(binary + ".srchash", "w") as f:
        f.write(digest + "\n")


HEADER = os.path.join(HERE, "..", "include", "sniffles_amd.h")


def build_fast(force: bool = False) -> str:
    """The CPython extension that turns record tables into SVCall objects (csrc/snf_pyfast.c; host-side formatting only)."""
    import sysconfig
    so, src = fast_so(), os.path.join(CSRC, "snf_pyfast.c")
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-pthread", "-I", sysconfig.get_paths()["include"], src, "-o", so]
    digest = _digest([src, HEADER], " ".join(cmd[:-3]))
    if force or _stale(so, digest):
        subprocess.run(cmd, check=True)
        _stamp(so, digest)
    return so


def _lib_digest() -> str:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "snf_pyfast.c"] + [HEADER]
    return _digest(deps, " ".join(FLAGS + SOURCES))


def needs_build() -> bool:
    return _stale(SO, _lib_digest())


def build(force: bool = False, verbose: bool = False) -> str:
    try:
        build_fast(force)
    except (OSError, subprocess.CalledProcessError) as e:      # no C compiler / Python.h: the pure-Python materialiser serves
        import warnings
        warnings.warn(f"sniffles_amd._snf_fast was not built ({e}); SVCall objects will be filled by the Python fallback")
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    digest = _lib_digest()
    subprocess.run(cmd, check=True)
    _stamp(SO, digest)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
