"""TEST-ONLY: the library the GPU-less tests run on.

There is ONE host tier: the product's four HIP translation units compiled UNCHANGED with g++ against the stand-in for the HIP
runtime in tests/emu/simt (every thread a fibre, cross-lane operations / DPP / barriers modelled; see tests/emu/simt.py).
This module is the name the tests have always imported it under (`emu.emu.lib()`); the former second tier - a `-DSNF_EMU`
build of serial-loop halves inside the product sources - is gone, and with it every `#ifdef SNF_EMU` in sniffles_amd/csrc.
Never shipped, never loadable through `sniffles_amd.lib.load()`.
"""
from . import simt as _simt

SO = _simt.SO


def build():
    return _simt.build()


def lib():
    """The host-tier library, bound; handing it out also makes it the library the product package works on (see simt.lib)."""
    return _simt.lib()
