"""Seam B4 of the reference: `consensus.novel_from_reads`.

`novel_from_reads(best_lead, other_leads, klen, skip, skip_repetitive, debug=False) -> str` has the signature and the
result of the reference function (`src/sniffles/consensus.py:280-394`; called from `postprocessing.py:63`): only the
`.seq` attribute of the leads is read.  The work is done by the workgroup consensus kernels of the library
(`snf_consensus_batch`, include/sniffles_amd.h); there is no CPU fallback.  `novel_from_reads_batch` runs many
independent problems in one launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def novel_from_reads_batch(problems, klen: int, device: int = 0):
    """problems: iterable of (best_seq, [other_seq, ...], skip[, skip_repetitive]) with str/bytes sequences -> list of str."""
    problems = list(problems)
    n = len(problems)
    if n == 0:
        return []
    enc = lambda s: s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")
    chunks, best_off, best_len, skips, skips_rep, o_index, o_off, o_len = [], [], [], [], [], [0], [], []
    pos = 0
    for best, others, skip, *rest in problems:
        b = enc(best)
        skips_rep.append(int(rest[0]) if rest else int(skip))
        best_off.append(pos); best_len.append(len(b)); skips.append(int(skip)); chunks.append(b); pos += len(b)
        for o in others:
            ob = enc(o)
            o_off.append(pos); o_len.append(len(ob)); chunks.append(ob); pos += len(ob)
        o_index.append(len(o_off))
    pool = np.frombuffer(b"".join(chunks) or b"\0", np.uint8)
    best_off = np.asarray(best_off, np.int64); best_len = np.asarray(best_len, np.int32); skips = np.asarray(skips, np.int32)
    skips_rep = np.asarray(skips_rep, np.int32)
    o_index = np.asarray(o_index, np.int64)
    o_off = np.asarray(o_off or [0], np.int64); o_len = np.asarray(o_len or [0], np.int32)
    out_off = np.zeros(n + 1, np.int64); out_off[1:] = np.cumsum(best_len)
    out = np.zeros(max(1, int(out_off[-1])), np.uint8)
    L = _lib.load()
    u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    rc = L.snf_consensus_batch(device, int(klen), pool.ctypes.data_as(u8p), C.c_int64(pos), C.c_int64(n),
                               best_off.ctypes.data_as(i64p), best_len.ctypes.data_as(i32p), skips.ctypes.data_as(i32p),
                               skips_rep.ctypes.data_as(i32p), o_index.ctypes.data_as(i64p), o_off.ctypes.data_as(i64p), o_len.ctypes.data_as(i32p),
                               out.ctypes.data_as(u8p), out_off.ctypes.data_as(i64p))
    if rc != 0:
        raise _lib.SnifflesAmdError(L.snf_last_error().decode())
    raw = out.tobytes()
    return [raw[int(out_off[i]):int(out_off[i + 1])].decode("latin-1") for i in range(n)]


def novel_from_reads(best_lead, other_leads, klen, skip, skip_repetitive, debug=False):
    """Drop-in for `sniffles.consensus.novel_from_reads` (reference consensus.py:280)."""
    return novel_from_reads_batch([(best_lead.seq, [ld.seq for ld in other_leads], skip, skip_repetitive)], klen)[0]
