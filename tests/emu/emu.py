"""TEST-ONLY host emulation of the HIP kernels.

Compiles sniffles_amd/csrc/*.hip with g++ -DSNF_EMU (every kernel body runs as a serial loop,
rocPRIM sort/scan replaced by std::) so the kernel LOGIC can be exercised in the GPU-less build
container.  Never shipped, never loadable through `sniffles_amd.lib.load()`.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sniffles_amd", "csrc")
SO = os.path.join(HERE, "_build", "libsnf_emu.so")
_lib = None


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in ("snf_lib.hip", "snf_myers.hip", "snf_combine.hip", "snf_extract.hip")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "sniffles_amd.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        cmd = ["g++", "-x", "c++", "-std=c++17", "-DSNF_EMU", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-Wno-misleading-indentation", "-Wno-unknown-pragmas"] + srcs + ["-o", SO]
        subprocess.run(cmd, check=True)
    return SO


def lib():
    global _lib
    if _lib is None:
        from sniffles_amd import lib as L
        _lib = L.bind(C.CDLL(build()))
    return _lib
