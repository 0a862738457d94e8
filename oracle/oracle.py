"""TEST INFRASTRUCTURE ONLY - loader for the C oracle (oracle/snf_oracle.c).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only.  The product package `sniffles_amd` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "snf_oracle.c")
SO = os.path.join(HERE, "_build", "libsnf_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    hdr = os.path.join(HERE, "..", "include", "sniffles_amd.h")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        # -ffp-contract=off: no fused multiply-add, results must match CPython's double arithmetic
        cmd = ["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
               "-Wall", "-Wextra", "-Wno-unused-parameter", "-o", SO, SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return SO


def lib():
    global _lib
    if _lib is None:
        from sniffles_amd import abi
        _lib = C.CDLL(build())
        _lib.snf_oracle_run.argtypes = [C.POINTER(abi.snf_config_t), C.POINTER(abi.snf_task_input_t), C.c_int,
                                        C.c_int, C.POINTER(C.c_void_p)]
        _lib.snf_oracle_run.restype = C.c_int
        _lib.snf_oracle_result.argtypes = [C.c_void_p]
        _lib.snf_oracle_result.restype = C.POINTER(abi.snf_result_t)
        _lib.snf_oracle_free.argtypes = [C.c_void_p]
        _lib.snf_oracle_free.restype = None
        _lib.snf_oracle_edit_distance.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        _lib.snf_oracle_edit_distance.restype = C.c_int64
        _lib.snf_oracle_edit_distance_myers.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        _lib.snf_oracle_edit_distance_myers.restype = C.c_int64
        _lib.snf_oracle_np_sum.argtypes = [C.POINTER(C.c_double), C.c_int64]
        _lib.snf_oracle_np_sum.restype = C.c_double
        _lib.snf_oracle_stdev.argtypes = [C.POINTER(C.c_int64), C.c_int64]
        _lib.snf_oracle_stdev.restype = C.c_double
        _lib.snf_oracle_hot_seconds.restype = C.c_double
        _lib.snf_oracle_combine_resolve.argtypes = [C.POINTER(abi.snf_config_t), C.POINTER(abi.snf_combine_problem_t)]
        _lib.snf_oracle_combine_resolve.restype = C.c_int
    return _lib


def run(cfg, tasks, finalize: bool = True):
    """Run the oracle on a list of TaskInput; returns sniffles_amd.abi.Result."""
    from sniffles_amd import abi
    L = lib()
    keep = []
    cs = abi.config_struct(cfg)
    arr = (abi.snf_task_input_t * len(tasks))(*[abi.task_struct(t, keep) for t in tasks])
    h = C.c_void_p()
    rc = L.snf_oracle_run(C.byref(cs), arr, len(tasks), int(finalize), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"snf_oracle_run failed: {rc}")
    try:
        return abi.Result(L.snf_oracle_result(h).contents)
    finally:
        L.snf_oracle_free(h)


def edit_distance(a: bytes, b: bytes) -> int:
    return int(lib().snf_oracle_edit_distance(a, len(a), b, len(b)))


def edit_distance_myers(a: bytes, b: bytes) -> int:
    """The same distance by the bit-parallel algorithm edlib implements (the edlib stand-in of the config-4 reference baseline)."""
    return int(lib().snf_oracle_edit_distance_myers(a, len(a), b, len(b)))


def np_sum(x: np.ndarray) -> float:
    x = np.ascontiguousarray(x, np.float64)
    return float(lib().snf_oracle_np_sum(x.ctypes.data_as(C.POINTER(C.c_double)), x.shape[0]))


def stdev(x) -> float:
    x = np.ascontiguousarray(x, np.int64)
    return float(lib().snf_oracle_stdev(x.ctypes.data_as(C.POINTER(C.c_int64)), x.shape[0]))


def hot_seconds() -> float:
    """Seconds the last run() spent in call_candidates + finalize_candidates (coverage-vector build excluded)."""
    return float(lib().snf_oracle_hot_seconds())


def combine_resolve(cfg, problem) -> None:
    """Group assignment of one packed resolve_block_groups problem (abi.combine_problem); fills its out_group."""
    from sniffles_amd import abi
    cs = abi.config_struct(cfg)
    rc = lib().snf_oracle_combine_resolve(C.byref(cs), C.byref(problem))
    if rc != 0:
        raise RuntimeError("snf_oracle_combine_resolve failed")
