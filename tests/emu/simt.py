"""TEST-ONLY: the gfx950 wave kernels on the host.

tests/emu/emu.py compiles the `SNF_EMU` halves of sniffles_amd/csrc (thread-per-item bodies as serial loops); the wave /
workgroup kernels (`snf_wave_*.h`) are outside it.  This module compiles them UNCHANGED with g++ against the fibre shim in
tests/emu/simt/hip/hip_runtime.h (every thread of a workgroup is a cooperative fibre; cross-lane operations, DPP and
barriers are modelled) so their logic can be checked against the reference's golden vectors without a GPU.  Never
shipped, never loadable through `sniffles_amd.lib.load()`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sniffles_amd", "csrc")
SIMT = os.path.join(HERE, "simt")
OBJ = os.path.join(HERE, "_build", "simt_obj")
SO = os.path.join(HERE, "_build", "libsnf_simt.so")
SO_UB = os.path.join(HERE, "_build", "libsnf_simt_ub.so")
# the product's four translation units, unchanged, plus the shim's counters and the consensus-instance harness
SOURCES = [os.path.join(CSRC, f) for f in ("snf_lib.hip", "snf_myers.hip", "snf_combine.hip", "snf_extract.hip")] + \
          [os.path.join(SIMT, f) for f in ("simt_api.cpp", "wave_cons_harness.cpp", "shim_selftest.cpp")]
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-fno-gnu-unique", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-misleading-indentation", "-Wno-unknown-pragmas", "-Wno-attributes", "-I" + SIMT, "-I" + CSRC]
_lib = None
_lib_ub = None


def build(sanitize=False):
    """sanitize: a second library with -fsanitize=undefined,bounds-strict (shifts by the operand width or more, signed overflow,
    indices beyond statically sized arrays - LDS arrays included - ...): what x86 tolerates silently the GPU may not."""
    so = SO_UB if sanitize else SO
    obj = OBJ + ("_ub" if sanitize else "")
    os.makedirs(obj, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "sniffles_amd.h")]
    for d, _, fs in os.walk(SIMT):
        deps += [os.path.join(d, f) for f in fs]
    def stale():
        return not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
    if stale():
        import fcntl
        with open(os.path.join(obj, ".lock"), "w") as lock:      # one builder at a time (pytest-xdist workers start together)
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                flags = list(FLAGS)
                if sanitize:
                    flags[flags.index("-O2")] = "-O1"
                    flags += ["-fsanitize=undefined", "-fsanitize=bounds-strict", "-fno-sanitize=alignment,vptr"]   # (unaligned 8 / 16-byte loads are the kernels' idiom)
                objs, procs = [], []
                for src in SOURCES:
                    o = os.path.join(obj, os.path.basename(src) + ".o")
                    objs.append(o)
                    procs.append(subprocess.Popen(["g++"] + flags + ["-c", src, "-o", o]))
                if any(p.wait() != 0 for p in procs):
                    raise RuntimeError("the SIMT build of sniffles_amd/csrc failed")
                tmp = so + ".tmp%d" % os.getpid()
                subprocess.run(["g++", "-shared", "-o", tmp] + (["-fsanitize=undefined"] if sanitize else []) + objs + ["-lpthread", "-ldl"], check=True)
                os.replace(tmp, so)      # (a reader never sees a half-written library)
    return so


def _bind(path):
    from sniffles_amd import lib as L
    h = L.bind(C.CDLL(path))
    u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    h.simt_consensus_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int64, C.c_int64, i64p, i32p, i32p,
                                       i64p, i64p, i32p, u8p, i64p, i32p, i64p]
    h.simt_consensus_batch.restype = C.c_int
    h.simt_last_error.restype = C.c_char_p
    h.snf_simt_counters.argtypes = [C.POINTER(C.c_ulonglong)]
    h.snf_simt_unmodelled.restype = C.c_ulonglong
    return h


def _activate(h):
    """Handing the host tier out makes it the library the product package works on (`sniffles_amd.lib.use_library`);
    tests/conftest.py selects the real one again before every test."""
    from sniffles_amd import lib as _plib
    _plib.use_library(h)


def lib_sanitized():
    global _lib_ub
    if _lib_ub is None:
        _lib_ub = _bind(build(sanitize=True))
    _activate(_lib_ub)
    return _lib_ub


def lib():
    """The WHOLE library (every kernel, wave kernels included) on the host: same C-ABI as libsniffles_amd.so."""
    global _lib
    if _lib is None:
        _lib = _bind(build())
    _activate(_lib)
    return _lib


def counters():
    c = (C.c_ulonglong * 9)()
    lib().snf_simt_counters(c)
    return dict(zip(("switches", "wave_ops", "block_syncs", "divergent_ops", "launches", "unmodelled", "lockstep_faults", "lockstep_merges",
                     "lockstep_conflicts"), [int(x) for x in c]))


def unmodelled():
    """Launches abandoned so far because they need lock-step execution (x_big: clusters of more than 64 leads)."""
    return int(lib().snf_simt_unmodelled())


def consensus_batch(problems, klen, mode=0, nw=4, grid_cap=0, min_reads=0, _lib=None):
    """problems: [(best, [others], skip)] -> (list of str, classes, handed_over).  Same packing as
    sniffles_amd.consensus.novel_from_reads_batch; the kernels are the product's, run through the fibre shim.
    mode 0: the class the product picks, 2: SMALL calls through the LARGE instance, 4: everything through ROWS;
    nw: waves per workgroup (SMALL: 4 or 1, LARGE: 4 or 8); grid_cap: workgroups stride over the calls;
    min_reads: calls with fewer other reads take the verbatim-copy kernel."""
    problems = list(problems)
    n = len(problems)
    enc = lambda s: s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")
    chunks, best_off, best_len, skips, o_index, o_off, o_len = [], [], [], [], [0], [], []
    pos = 0
    for best, others, skip in problems:
        b = enc(best)
        best_off.append(pos); best_len.append(len(b)); skips.append(int(skip)); chunks.append(b); pos += len(b)
        for o in others:
            ob = enc(o)
            o_off.append(pos); o_len.append(len(ob)); chunks.append(ob); pos += len(ob)
        o_index.append(len(o_off))
    pool = np.frombuffer(b"".join(chunks) or b"\0", np.uint8)
    best_off = np.asarray(best_off, np.int64); best_len = np.asarray(best_len, np.int32); skips = np.asarray(skips, np.int32)
    o_index = np.asarray(o_index, np.int64)
    o_off = np.asarray(o_off or [0], np.int64); o_len = np.asarray(o_len or [0], np.int32)
    out_off = np.zeros(n + 1, np.int64); out_off[1:] = np.cumsum(best_len)
    out = np.zeros(max(1, int(out_off[-1])), np.uint8)
    cls = np.zeros(max(1, n), np.int32)
    handed = C.c_int64(0)
    L = _lib or lib()
    u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    rc = L.simt_consensus_batch(mode, nw, grid_cap, min_reads, int(klen), pool.ctypes.data_as(u8p), C.c_int64(pos), C.c_int64(n),
                                best_off.ctypes.data_as(i64p), best_len.ctypes.data_as(i32p), skips.ctypes.data_as(i32p),
                                o_index.ctypes.data_as(i64p), o_off.ctypes.data_as(i64p), o_len.ctypes.data_as(i32p),
                                out.ctypes.data_as(u8p), out_off.ctypes.data_as(i64p), cls.ctypes.data_as(i32p), C.byref(handed))
    if rc != 0:
        raise RuntimeError(L.simt_last_error().decode())
    raw = out.tobytes()
    return [raw[int(out_off[i]):int(out_off[i + 1])].decode("latin-1") for i in range(n)], cls[:n].tolist(), handed.value
