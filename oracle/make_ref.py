"""TEST INFRASTRUCTURE ONLY - build recipe for `oracle/_ref/`: the UNMODIFIED reference as a compiled artefact that travels.

The reference is pure Python, so "compiling it from the sources where they lie" is byte-compiling: every module of
`/root/reference/src/sniffles/*.py` is compiled by THIS interpreter (`py_compile`, unchecked-hash pycs) into
`oracle/_ref/sniffles/<module>.pyc` - the sourceless layout CPython imports directly.  No reference source text is
written anywhere in the repository; `oracle/_ref/` is git-ignored (it stays out of history) but NOT gpurun-ignored, so
it reaches the GPU box next to the built `.so` files.  `__graft_entry__.build()` runs this whenever `/root/reference`
is present (the build container); on the GPU box the prebuilt directory is only used.

Consumers (all of them checkers / baselines, never the product path):
  * `oracle/ref_harness.py::load_reference` falls back to `oracle/_ref` when `/root/reference/src` is absent;
  * `bench.py`'s `cpu_baseline` leg times the reference's own `Task.call_candidates + finalize_candidates` through it
    (`cpu_baseline.kind = "reference"`, `oracle/ref_pool.py`);
  * `tests/test_insitu_seam.py`'s GPU cases (the reference's `CallTask.execute` around the HIP library on the MI355X).

A pyc is tied to the interpreter's bytecode magic: MANIFEST.json records it and `ref_root()` refuses a mismatch
(the GPU box runs the same image, Python 3.10.12).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.environ.get("SNF_REFERENCE_SRC", "/root/reference/src")   # (the variable exists to test the staged form in the build container)
OUT_ROOT = os.path.join(HERE, "_ref")
PKG = "sniffles"
MANIFEST = os.path.join(OUT_ROOT, "MANIFEST.json")


def _magic() -> str:
    return importlib.util.MAGIC_NUMBER.hex()


def _sources() -> list:
    """(relative module path, absolute source path) of the reference package, sub-packages included."""
    out = []
    base = os.path.join(SRC_ROOT, PKG)
    for d, _dirs, files in os.walk(base):
        for f in sorted(files):
            if f.endswith(".py"):
                p = os.path.join(d, f)
                out.append((os.path.relpath(p, SRC_ROOT), p))
    return sorted(out)


def source_available() -> bool:
    return os.path.isfile(os.path.join(SRC_ROOT, PKG, "__init__.py"))


def _read_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def staged_ok() -> bool:
    """A usable compiled reference is present (right bytecode magic, every listed module on disk)."""
    m = _read_manifest()
    if not m or m.get("magic") != _magic():
        return False
    return all(os.path.isfile(os.path.join(OUT_ROOT, rel + "c")) for rel in m.get("modules", {}))


def build(force: bool = False):
    """Compile the reference into oracle/_ref/ (build container only).  Returns the directory, or None if there is neither a
    reference checkout nor a previously staged build."""
    if not source_available():
        return OUT_ROOT if staged_ok() else None
    srcs = _sources()
    digests = {}
    for rel, p in srcs:
        with open(p, "rb") as f:
            digests[rel] = hashlib.sha256(f.read()).hexdigest()
    m = _read_manifest()
    if not force and m and m.get("magic") == _magic() and m.get("modules") == digests and staged_ok():
        return OUT_ROOT
    for rel, p in srcs:
        dst = os.path.join(OUT_ROOT, rel + "c")             # sniffles/cluster.py -> _ref/sniffles/cluster.pyc (sourceless import layout)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(p, cfile=dst, dfile=os.path.join("<reference>", rel), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    version = None
    try:
        with open(os.path.join(SRC_ROOT, PKG, "__init__.py")) as f:
            for line in f:
                if "version" in line.lower() and "=" in line:
                    version = line.split("=", 1)[1].strip().strip("\"'")
    except OSError:
        pass
    with open(MANIFEST, "w") as f:
        json.dump(dict(what="byte-compiled, unmodified fritzsedlazeck/Sniffles src/sniffles (oracle/make_ref.py); sha256 of each source",
                       python=sys.version.split()[0], magic=_magic(), reference_version=version, modules=digests), f, indent=1)
    return OUT_ROOT


def ref_root():
    """Directory to put on sys.path to import the reference package: the checkout if present, else the staged build."""
    if source_available():
        return SRC_ROOT
    if staged_ok():
        return OUT_ROOT
    return None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
