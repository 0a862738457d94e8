"""ctypes mirror of include/sniffles_amd.h (struct layouts, enums, config builder).

This is the boundary definition only - no compute.  Used by the product binding
(`sniffles_amd.lib`) and, in tests, by the oracle loader (`oracle/oracle.py`).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .soa import LEAD_FIELDS, TaskInput

ABI_VERSION = 5

FILTERS = [
    "PASS", "STDEV_POS", "STDEV_LEN", "SINGLE_BREAK", "SVLEN_MIN", "STRAND_BND", "COV_CHANGE_DEL",
    "COV_CHANGE_DUP", "COV_CHANGE_INS", "INLINE_SA", "COV_VAR", "COV_CHANGE_FRAC_US", "COV_CHANGE_FRAC_SC",
    "COV_CHANGE_FRAC_CE", "COV_CHANGE_FRAC_ED", "SUPPORT_MIN", "GT_FAILED", "GT", "COV_MIN_GT", "ALN_NM",
    "MOSAIC_VAF", "SVLEN_MAX_MOSAIC", "STRAND", "STRAND_MOSAIC", "SVLEN_MIN_MOSAIC", "COV_MIN",
    "NOT_MOSAIC_VAF", "MOSAIC_SV_CLOSE_EDGE",
]
TASK_OK, TASK_ERR_UNBOUND_END = 0, 1

i32, i64, f64, u8p = C.c_int32, C.c_int64, C.c_double, C.POINTER(C.c_uint8)


class snf_config_t(C.Structure):
    _fields_ = [
        ("cluster_binsize", i32), ("cluster_merge_pos", i32), ("cluster_merge_bnd", i32),
        ("cluster_resplit_binsize", i32),
        ("cluster_r", f64), ("cluster_repeat_h", f64), ("cluster_repeat_h_max", f64), ("cluster_merge_len", f64),
        ("minsvlen", i32), ("minsvlen_screen", i32), ("minsvlen_hard_cap", i32), ("minsupport", i32),
        ("minsupport_auto_base", f64), ("minsupport_auto_mult", f64),
        ("minsupport_auto_regional_coverage_weight", f64),
        ("long_ins_length", i32), ("long_del_length", i32), ("long_dup_length", i32), ("long_inv_length", i32),
        ("long_ins_rescale_base", f64), ("long_ins_rescale_mult", f64),
        ("long_del_coverage", f64), ("long_dup_coverage", f64),
        ("dev_longer_del", i32), ("dev_longer_dup", i32),
        ("consensus_max_reads_bin", i32), ("consensus_min_reads", i32), ("consensus_kmer_len", i32),
        ("consensus_kmer_skip_base", i32), ("consensus_kmer_skip_seqlen_mult", f64),
        ("precise", i32), ("coverage_binsize", i32), ("coverage_updown_bins", i32),
        ("genotype_ploidy", i32), ("genotype_min_z_score", i32), ("genotype_error", f64),
        ("qc_stdev", i32), ("qc_stdev_abs_max", i32), ("qc_strand", i32), ("qc_coverage", i32),
        ("qc_bnd_filter_strand", i32), ("qc_nm", i32), ("qc_nm_measure", i32), ("pass_only", i32),
        ("qc_coverage_max_change_frac", f64), ("qc_nm_mult", f64), ("dev_inline_sa_support_max", f64),
        ("dev_min_dup_vaf", f64),
        ("mosaic", i32), ("mosaic_min_reads", i32), ("mosaic_use_strand_thresholds", i32),
        ("max_svlen_mosaic", i32), ("mosaic_qc_invdup_min_length", i32), ("mosaic_qc_nm", i32),
        ("mosaic_qc_strand", i32), ("mosaic_include_germline", i32),
        ("mosaic_af_max", f64), ("mosaic_af_min", f64),
        ("dev_min_close_edge_dist", i32), ("dev_minreads_extra", i32), ("dev_maxsvlen_extra", i32), ("_pad0", i32),
        ("dev_min_read_close_edge_prop", f64),
        ("dev_min_leads_cluster", i32), ("repeat", i32), ("phase", i32), ("detect_large_ins", i32),
        ("no_consensus", i32), ("symbolic", i32), ("dev_no_resplit", i32), ("dev_no_resplit_repeat", i32),
        ("dev_output_candidates", i32), ("mode_call_sample", i32),
        ("phase_conflict_threshold", f64),
        ("combine_match", i32), ("combine_match_max", i32), ("combine_separate_intra", i32), ("_pad1", i32),
        ("combine_pctseq", f64),
        ("no_qc", i32), ("sort", i32),
    ]


_CT = {np.dtype(np.int32): C.c_int32, np.dtype(np.uint32): C.c_uint32, np.dtype(np.int64): C.c_int64,
       np.dtype(np.float64): C.c_double, np.dtype(np.uint8): C.c_uint8}

# order of pointer members in snf_task_input_t after n_leads
_LEAD_PTR_ORDER = ["ref_start", "ref_end", "qry_start", "qry_end", "svlen", "read_len", "qname_id", "read_id",
                   "ps_rank", "mate_contig", "mate_ref_start", "seq_len", "seq_off", "nm", "svtype", "strand",
                   "mapq", "source", "hap", "is_sa", "bnd_is_first", "bnd_is_reverse"]
_LEAD_DT = dict(LEAD_FIELDS)


class snf_task_input_t(C.Structure):
    _fields_ = ([("task_id", i32), ("sv_id_start", i32), ("contig_len", i32), ("ps_null_rank", i32),
                 ("qc_nm_threshold", f64), ("n_leads", i64)] +
                [(n, C.POINTER(_CT[np.dtype(_LEAD_DT[n])])) for n in _LEAD_PTR_ORDER] +
                [("seq_pool_len", i64), ("seq_pool", u8p),
                 ("n_reads", i64), ("read_start", C.POINTER(C.c_int32)), ("read_end", C.POINTER(C.c_int32)),
                 ("read_hp", u8p),
                 ("n_tr", i64), ("tr_start", C.POINTER(C.c_int32)), ("tr_end", C.POINTER(C.c_int32)),
                 ("n_nmask", i64), ("nmask_start", C.POINTER(C.c_int32)), ("nmask_end", C.POINTER(C.c_int32))])


class snf_call_t(C.Structure):
    _fields_ = [
        ("task_index", i32), ("sv_id", i32), ("svtype", i32), ("pos", i32), ("end", i32), ("svlen", i32),
        ("support", i32), ("support_long", i32), ("support_sa", i32), ("qual", i32), ("precise", i32),
        ("fwd", i32), ("rev", i32), ("qc", i32), ("filter", i32), ("cov", i32 * 5),
        ("sa_count", i32), ("n_leads", i32), ("sa_frac", f64), ("nm", f64), ("stdev_pos", f64), ("stdev_len", f64),
        ("mate_contig", i32), ("mate_ref_start", i32), ("bnd_is_first", i32), ("bnd_is_reverse", i32),
        ("gt_set", i32), ("gt_a", i32), ("gt_b", i32), ("gt_gq", i32), ("gt_dr", i32), ("gt_dv", i32),
        ("gt_hp", i32), ("gt_ps", i32), ("vaf", f64),
        ("ph_set", i32), ("ph_hp", i32), ("ph_ps", i32), ("ph_hp_support", i32), ("ph_ps_support", i32),
        ("ph_hp_pass", i32), ("ph_ps_pass", i32),
        ("alt_len", i32), ("alt_off", i64), ("rn_off", i64), ("rn_len", i32),
        ("cluster_start", i32), ("cluster_end", i32), ("cluster_seed_index", i32),
    ]


CALL_DTYPE = np.dtype(snf_call_t)


class snf_result_t(C.Structure):
    _fields_ = [
        ("n_calls", i64), ("calls", C.POINTER(snf_call_t)),
        ("alt_pool_len", i64), ("alt_pool", u8p),
        ("rnames_len", i64), ("rnames", C.POINTER(C.c_uint32)),
        ("n_tasks", i64), ("task_status", C.POINTER(C.c_int32)),
        ("task_call_off", C.POINTER(C.c_int64)), ("coverage_average_total", C.POINTER(C.c_double)),
    ]


class snf_export_layout_t(C.Structure):
    _fields_ = [("n_calls", i64), ("rnames_len", i64), ("alt_pool_len", i64), ("off_rnames", i64), ("off_alt", i64), ("bytes", i64)]


# enum snf_output (snf_batch_set_output)
OUT_CANDIDATES, OUT_EXECUTE, OUT_DEVICE = 0, 1, 2


class snf_clusters_t(C.Structure):
    _fields_ = [("n_clusters", i64)] + [(n, C.POINTER(C.c_int32)) for n in
                                        ("task_index", "svtype", "start", "end", "seed", "seed_index", "n_leads_long")] + \
               [("repeat", u8p), ("lead_off", C.POINTER(C.c_int64)), ("n_leads", i64), ("lead", C.POINTER(C.c_int32)),
                ("lead_svlen", C.POINTER(C.c_int32))]


class snf_combine_problem_t(C.Structure):
    _fields_ = [
        ("svtype", i32), ("n_cands", i32), ("n_groups", i32), ("n_sample_ids", i32),
        ("pos", C.POINTER(C.c_int32)), ("svlen", C.POINTER(C.c_int32)), ("support", C.POINTER(C.c_int32)),
        ("sample_id", C.POINTER(C.c_int32)), ("mate_contig", C.POINTER(C.c_int32)), ("mate_ref_start", C.POINTER(C.c_int32)),
        ("alt_off", C.POINTER(C.c_int64)), ("alt_pool", u8p),
        ("g_pos_mean", C.POINTER(C.c_double)), ("g_len_mean", C.POINTER(C.c_double)), ("g_mate_mean", C.POINTER(C.c_double)),
        ("g_size", C.POINTER(C.c_int32)), ("g_mate_contig", C.POINTER(C.c_int32)),
        ("g_alt_off", C.POINTER(C.c_int64)), ("g_alt_pool", u8p),
        ("g_samples_off", C.POINTER(C.c_int64)), ("g_samples", C.POINTER(C.c_int32)),
        ("out_group", C.POINTER(C.c_int32)),
        ("n_windows", i32), ("win_off", C.POINTER(C.c_int32)), ("win_bin", C.POINTER(C.c_int32)), ("win_thr", C.POINTER(C.c_double)),
    ]


def combine_problem(svtype_code: int, cands: dict, groups: dict, n_sample_ids: int, keep: list, windows=None):
    """Pack one resolve_block_groups call.  `cands`: pos, svlen, support, sample_id, mate_contig, mate_ref_start
    (int lists) and alts (list of bytes); `groups`: pos_mean, len_mean, mate_mean, size, mate_contig, alts, samples
    (list of lists).  Returns (struct, out_group numpy array)."""
    q = snf_combine_problem_t()
    n, g = len(cands["pos"]), len(groups["pos_mean"])
    q.svtype, q.n_cands, q.n_groups, q.n_sample_ids = svtype_code, n, g, max(1, n_sample_ids)

    def arr(x, dt):
        a = np.ascontiguousarray(np.asarray(x, dtype=dt).reshape(-1))
        if a.size == 0:
            a = np.zeros(1, dt)
        keep.append(a)
        return a

    def pool(strs):
        off = np.zeros(len(strs) + 1, np.int64)
        for i, s in enumerate(strs):
            off[i + 1] = off[i] + len(s)
        data = np.frombuffer(b"".join(strs) + b"\0", np.uint8).copy()
        keep.extend([off, data])
        return off, data

    for name in ("pos", "svlen", "support", "sample_id", "mate_contig", "mate_ref_start"):
        setattr(q, name, _ptr(arr(cands[name], np.int32), C.c_int32))
    off, data = pool(cands["alts"])
    q.alt_off, q.alt_pool = _ptr(off, C.c_int64), _ptr(data, C.c_uint8)
    q.g_pos_mean = _ptr(arr(groups["pos_mean"], np.float64), C.c_double)
    q.g_len_mean = _ptr(arr(groups["len_mean"], np.float64), C.c_double)
    q.g_mate_mean = _ptr(arr(groups["mate_mean"], np.float64), C.c_double)
    q.g_size = _ptr(arr(groups["size"], np.int32), C.c_int32)
    q.g_mate_contig = _ptr(arr(groups["mate_contig"], np.int32), C.c_int32)
    off, data = pool(groups["alts"])
    q.g_alt_off, q.g_alt_pool = _ptr(off, C.c_int64), _ptr(data, C.c_uint8)
    soff = np.zeros(g + 1, np.int64)
    flat = []
    for i, ss in enumerate(groups["samples"]):
        flat.extend(sorted(ss))
        soff[i + 1] = len(flat)
    keep.append(soff)
    q.g_samples_off = _ptr(soff, C.c_int64)
    q.g_samples = _ptr(arr(flat, np.int32), C.c_int32)
    out = np.full(max(n, 1), -1, np.int32)
    keep.append(out)
    q.out_group = _ptr(out, C.c_int32)
    if windows is not None:   # (win_off [n+1], win_bin [n], win_thr [n]): chain of flush windows
        woff, wbin, wthr = windows
        q.n_windows = len(wbin)
        q.win_off = _ptr(arr(woff, np.int32), C.c_int32)
        q.win_bin = _ptr(arr(wbin, np.int32), C.c_int32)
        q.win_thr = _ptr(arr(wthr, np.float64), C.c_double)
    else:
        q.n_windows = 0
    return q, out


# numpy mirror of snf_combine_problem_t (pointers as addresses) for packing many problems without per-problem ctypes work
COMBINE_PROBLEM_DTYPE = np.dtype(
    [("svtype", np.int32), ("n_cands", np.int32), ("n_groups", np.int32), ("n_sample_ids", np.int32)] +
    [(n, np.uint64) for n in ("pos", "svlen", "support", "sample_id", "mate_contig", "mate_ref_start", "alt_off", "alt_pool",
                              "g_pos_mean", "g_len_mean", "g_mate_mean", "g_size", "g_mate_contig", "g_alt_off", "g_alt_pool",
                              "g_samples_off", "g_samples", "out_group")] +
    [("n_windows", np.int32), ("win_off", np.uint64), ("win_bin", np.uint64), ("win_thr", np.uint64)], align=True)
assert COMBINE_PROBLEM_DTYPE.itemsize == C.sizeof(snf_combine_problem_t)
assert all(COMBINE_PROBLEM_DTYPE.fields[n][1] == getattr(snf_combine_problem_t, n).offset for n, _ in snf_combine_problem_t._fields_)


def combine_chain_problems(svtype_codes, cand_lo, cand_hi, win_lo, win_hi, cols: dict, alts, win_off, win_bin, win_thr,
                           n_sample_ids: int, keep: list):
    """Pack P sub-chains (no initial groups) that are contiguous slices of shared candidate / window tables.

    svtype_codes, cand_lo, cand_hi, win_lo, win_hi: per sub-chain - its SV type, candidate range [lo, hi) in the shared
    columns `cols` (pos, svlen, support, sample_id, mate_contig, mate_ref_start: int sequences over ALL candidates) and
    `alts` (bytes per candidate), window range [lo, hi) in win_bin / win_thr; `win_off[w]` = first candidate of window w in
    the shared numbering, plus one padding entry at the end of the table.  Every struct points into
    the shared arrays, so the cost per sub-chain is a row of a numpy table.  Returns (ctypes struct array, out_group array)."""
    n_p = len(svtype_codes)
    if isinstance(alts, tuple):          # (offsets int64[n + 1], pool bytes): the columnar store hands the pool over as it is
        aoff = np.ascontiguousarray(alts[0], np.int64)
        n_c = len(aoff) - 1
        pool = alts[1] if isinstance(alts[1], np.ndarray) else np.frombuffer(alts[1] or b"\0", np.uint8)
    else:
        n_c = len(alts)
        aoff = np.zeros(n_c + 1, np.int64)
        if n_c:
            np.cumsum(np.fromiter((len(a) for a in alts), np.int64, n_c), out=aoff[1:])
        pool = np.frombuffer(b"".join(alts) + b"\0", np.uint8).copy()
    a32 = {k: np.ascontiguousarray(np.asarray(v, np.int32).reshape(-1)) if n_c else np.zeros(1, np.int32) for k, v in cols.items()}
    out = np.full(max(n_c, 1), -1, np.int32)
    cand_lo, cand_hi = np.asarray(cand_lo, np.int64), np.asarray(cand_hi, np.int64)
    win_lo, win_hi = np.asarray(win_lo, np.int64), np.asarray(win_hi, np.int64)
    # window offsets relative to the sub-chain's first candidate: nw + 1 entries per sub-chain
    nw = win_hi - win_lo
    wstart = np.zeros(n_p + 1, np.int64)
    np.cumsum(nw + 1, out=wstart[1:])
    idx = np.arange(int(wstart[-1]), dtype=np.int64) - np.repeat(wstart[:-1], nw + 1) + np.repeat(win_lo, nw + 1)
    woff_all = np.asarray(win_off, np.int64)
    rel = (woff_all[idx] - np.repeat(cand_lo, nw + 1)).astype(np.int32)
    # the closing entry of a sub-chain is its candidate count
    rel[wstart[1:] - 1] = (cand_hi - cand_lo).astype(np.int32)
    wbin = np.ascontiguousarray(np.asarray(win_bin, np.int32).reshape(-1)) if len(win_bin) else np.zeros(1, np.int32)
    wthr = np.ascontiguousarray(np.asarray(win_thr, np.float64).reshape(-1)) if len(win_thr) else np.zeros(1, np.float64)
    dz, iz, lz = np.zeros(1, np.float64), np.zeros(1, np.int32), np.zeros(2, np.int64)
    rec = np.zeros(n_p, COMBINE_PROBLEM_DTYPE)
    rec["svtype"] = np.asarray(svtype_codes, np.int32)
    rec["n_cands"] = (cand_hi - cand_lo).astype(np.int32)
    rec["n_sample_ids"] = max(1, int(n_sample_ids))
    for k in ("pos", "svlen", "support", "sample_id", "mate_contig", "mate_ref_start"):
        rec[k] = a32[k].ctypes.data + 4 * cand_lo.astype(np.uint64)
    rec["alt_off"] = aoff.ctypes.data + 8 * cand_lo.astype(np.uint64)      # absolute offsets into the shared pool
    rec["alt_pool"] = pool.ctypes.data
    for k in ("g_pos_mean", "g_len_mean", "g_mate_mean"):
        rec[k] = dz.ctypes.data
    rec["g_size"] = rec["g_mate_contig"] = rec["g_samples"] = iz.ctypes.data
    rec["g_alt_off"] = rec["g_samples_off"] = lz.ctypes.data
    rec["g_alt_pool"] = pool.ctypes.data
    rec["out_group"] = out.ctypes.data + 4 * cand_lo.astype(np.uint64)
    rec["n_windows"] = nw.astype(np.int32)
    rec["win_off"] = rel.ctypes.data + 4 * wstart[:-1].astype(np.uint64)
    rec["win_bin"] = wbin.ctypes.data + 4 * win_lo.astype(np.uint64)
    rec["win_thr"] = wthr.ctypes.data + 8 * win_lo.astype(np.uint64)
    keep.extend([a32, aoff, pool, out, rel, wbin, wthr, dz, iz, lz, rec])
    return (snf_combine_problem_t * n_p).from_buffer(rec), out


def config_struct(cfg) -> snf_config_t:
    """Build snf_config_t from a SnifflesConfig-compatible namespace (reference config.py:103-619)."""
    s = snf_config_t()
    g = lambda n, d=None: getattr(cfg, n, d)  # noqa: E731
    for name, ctype in snf_config_t._fields_:
        if name.startswith("_pad"):
            continue
        if name == "minsupport":
            v = g("minsupport")
            s.minsupport = -1 if v == "auto" else int(v)
            continue
        if name == "mode_call_sample":
            s.mode_call_sample = int(g("mode", "call_sample") == "call_sample")
            continue
        if name == "dev_output_candidates":
            s.dev_output_candidates = int(bool(g("dev_output_candidates", None)))
            continue
        v = g(name)
        if v is None:
            raise AttributeError(f"config is missing hot-path constant {name!r}")
        setattr(s, name, float(v) if ctype is f64 else int(v))
    if getattr(cfg, "dev_filter", False):
        raise NotImplementedError("--dev-filter (multi-filter strings) is a developer mode outside the hot path")
    return s


def _ptr(a: np.ndarray, ctype):
    return C.cast(a.__array_interface__["data"][0], C.POINTER(ctype))     # (a.ctypes.data_as is ~4 us per array)


def task_struct(ti: TaskInput, keep: list) -> snf_task_input_t:
    """Borrow the numpy buffers of `ti` into a snf_task_input_t (arrays appended to `keep` stay alive)."""
    ti.check_layout()      # values are validated by the library while it stages the columns
    t = snf_task_input_t()
    t.task_id, t.sv_id_start, t.contig_len = ti.task_id, ti.sv_id_start, ti.contig_len
    null_rank = -1
    if ti.ps_names is not None and "NULL" in ti.ps_names:
        null_rank = ti.ps_names.index("NULL")
    t.ps_null_rank = null_rank
    t.qc_nm_threshold = float(ti.qc_nm_threshold)
    t.n_leads = ti.n_leads
    for n in _LEAD_PTR_ORDER:
        a = ti.leads[n]
        keep.append(a)
        setattr(t, n, _ptr(a, _CT[a.dtype]))
    keep.append(ti.seq_pool)
    t.seq_pool_len = int(ti.seq_pool.shape[0])
    t.seq_pool = _ptr(ti.seq_pool, C.c_uint8)
    t.n_reads = ti.n_reads
    for n in ("read_start", "read_end"):
        a = getattr(ti, n)
        keep.append(a)
        setattr(t, n, _ptr(a, C.c_int32))
    keep.append(ti.read_hp)
    t.read_hp = _ptr(ti.read_hp, C.c_uint8)
    if ti.tr_start is None:
        t.n_tr = -1
    else:
        ts = np.ascontiguousarray(ti.tr_start, np.int32)
        te = np.ascontiguousarray(ti.tr_end, np.int32)
        keep += [ts, te]
        t.n_tr = int(ts.shape[0])
        t.tr_start, t.tr_end = _ptr(ts, C.c_int32), _ptr(te, C.c_int32)
    _nmask(t, ti, keep)
    return t


def _nmask(t, ti, keep: list) -> None:
    ns, ne = getattr(ti, "nmask_start", None), getattr(ti, "nmask_end", None)
    t.n_nmask = 0
    if ns is not None and len(ns):
        ns, ne = np.ascontiguousarray(ns, np.int32), np.ascontiguousarray(ne, np.int32)
        keep += [ns, ne]
        t.n_nmask = int(ns.shape[0])
        t.nmask_start, t.nmask_end = _ptr(ns, C.c_int32), _ptr(ne, C.c_int32)


def task_meta_struct(ti, keep: list) -> snf_task_input_t:
    """What snf_batch_add_task_device reads from the host: ids, contig length and the tandem repeats of a task whose columns
    are in HBM."""
    t = snf_task_input_t()
    t.task_id, t.sv_id_start, t.contig_len = ti.task_id, ti.sv_id_start, ti.contig_len
    if ti.tr_start is None:
        t.n_tr = -1
    else:
        ts = np.ascontiguousarray(ti.tr_start, np.int32)
        te = np.ascontiguousarray(ti.tr_end, np.int32)
        keep += [ts, te]
        t.n_tr = int(ts.shape[0])
        t.tr_start, t.tr_end = _ptr(ts, C.c_int32), _ptr(te, C.c_int32)
    _nmask(t, ti, keep)
    return t


class Result:
    """Host copy of a snf_result_t (numpy views copied out of library-owned memory)."""

    def __init__(self, r: snf_result_t, copy: bool = True):
        """copy=False: the arrays are views of the library's own (pinned) result buffers - valid until the next call on the
        batch; for callers that turn them into objects right away (Task.call_candidates / finalize_candidates)."""
        own = (lambda a: a.copy()) if copy else (lambda a: a)
        n = int(r.n_calls)
        self.calls = own(np.ctypeslib.as_array(C.cast(r.calls, C.POINTER(C.c_uint8)),
                                               shape=(n * CALL_DTYPE.itemsize,)).view(CALL_DTYPE)) if n else np.zeros(0, CALL_DTYPE)
        na = int(r.alt_pool_len)
        self.alt_pool = own(np.ctypeslib.as_array(r.alt_pool, shape=(na,))) if na else np.zeros(0, np.uint8)
        nr = int(r.rnames_len)
        self.rnames = own(np.ctypeslib.as_array(r.rnames, shape=(nr,))) if nr else np.zeros(0, np.uint32)
        nt = int(r.n_tasks)
        self.task_status = np.ctypeslib.as_array(r.task_status, shape=(nt,)).copy()
        self.task_call_off = np.ctypeslib.as_array(r.task_call_off, shape=(nt + 1,)).copy()
        self.coverage_average_total = np.ctypeslib.as_array(r.coverage_average_total, shape=(nt,)).copy()

    def alt(self, i: int):
        c = self.calls[i]
        if c["alt_len"] < 0:
            return None
        o = int(c["alt_off"])
        return self.alt_pool[o:o + int(c["alt_len"])].tobytes().decode("latin-1")

    def rn(self, i: int) -> np.ndarray:
        c = self.calls[i]
        o = int(c["rn_off"])
        return self.rnames[o:o + int(c["rn_len"])]


def none_if_nan(x):
    x = float(x)
    return None if math.isnan(x) else x


# ---------------------------------------------------------------------------------------------- signature extraction
class snf_extract_config_t(C.Structure):
    _fields_ = [("mapq", i32), ("min_alignment_length", i32), ("exclude_flags", i32), ("minsvlen_screen", i32),
                ("long_ins_length", i32), ("dev_seq_cache_maxlen", i32), ("max_splits_base", i32),
                ("detect_large_ins", i32), ("advanced_tags", i32), ("dev_keep_lowqual_splits", i32),
                ("max_splits_kb", f64)]


class snf_extract_input_t(C.Structure):
    _fields_ = [("records", u8p), ("records_len", i64), ("rec_off", C.POINTER(C.c_int64)), ("n_records", i64),
                ("qname_rank", C.POINTER(C.c_uint32)), ("region_ref_id", i32), ("region_rank", i32),
                ("region_start", i32), ("region_end", i32), ("read_id_offset", C.c_uint32), ("n_contigs", i32),
                ("contig_hash", C.POINTER(C.c_uint64)), ("contig_rank", C.POINTER(C.c_int32))]


class snf_extract_result_t(C.Structure):
    _fields_ = [("task", snf_task_input_t), ("n_ps", i64), ("ps_value", C.POINTER(C.c_int64)),
                ("read_id", C.c_uint32), ("read_count", i64), ("ms_count", C.c_float), ("ms_emit", C.c_float),
                ("algo_bytes", i64)]


def extract_config_struct(cfg) -> snf_extract_config_t:
    """`cfg`: anything with the SnifflesConfig attribute names extraction reads (config.py:190-215, 507-617)."""
    g = lambda name, default: getattr(cfg, name, default)
    ex = g("exclude_flags", None)
    qc_nm_measure = g("qc_nm_measure", g("qc_nm", True))
    return snf_extract_config_t(
        mapq=int(g("mapq", 20)), min_alignment_length=int(g("min_alignment_length", 1000)),
        exclude_flags=-1 if ex is None else int(ex), minsvlen_screen=int(g("minsvlen_screen", 45)),
        long_ins_length=int(g("long_ins_length", 2500)), dev_seq_cache_maxlen=int(g("dev_seq_cache_maxlen", 50000)),
        max_splits_base=int(g("max_splits_base", 3)), detect_large_ins=int(bool(g("detect_large_ins", True))),
        advanced_tags=int(bool(g("advanced_tags", qc_nm_measure or g("phase", False)))),
        dev_keep_lowqual_splits=int(bool(g("dev_keep_lowqual_splits", False))),
        max_splits_kb=float(g("max_splits_kb", 0.1)))


# ---- columnar candidate store of the multi-sample combine (snf_combine_call_groups)
NONE_I32 = -2**31
GROUP_CAND_DTYPE = np.dtype([("pos", "<i4"), ("svlen", "<i4"), ("end", "<i4"), ("support", "<i4"), ("qual", "<i4"), ("fwd", "<i4"),
                             ("rev", "<i4"), ("cov", "<i4", (5,)), ("gq", "<i4"), ("dr", "<i4"), ("dv", "<i4"), ("sample", "<i4"),
                             ("alt_len", "<i4"), ("gt_a", "i1"), ("gt_b", "i1"), ("qc", "u1"), ("pass", "u1"), ("precise", "u1"),
                             ("is_ins", "u1"), ("_pad", "u1", (2,))])
GROUP_OUT_DTYPE = np.dtype([("flush_win", "<i4"), ("emit", "<i4"), ("n_pass", "<i4"), ("n_present", "<i4"), ("pos", "<i4"),
                            ("svlen", "<i4"), ("end", "<i4"), ("alt_member", "<i4"), ("qual", "<i4"), ("support", "<i4"),
                            ("fwd", "<i4"), ("rev", "<i4"), ("cov", "<i4", (5,)), ("precise", "<i4"), ("n", "<i4"), ("_pad", "<i4"),
                            ("stdev_pos", "<f8"), ("stdev_len", "<f8")])
assert GROUP_CAND_DTYPE.itemsize == 76 and GROUP_OUT_DTYPE.itemsize == 96


class snf_group_call_config_t(C.Structure):
    _fields_ = [("n_samples", i32), ("no_qc", i32), ("combine_low_confidence_abs", i32), ("combine_output_filtered", i32),
                ("dev_combine_medians", i32), ("minsvlen_screen", i32), ("combine_high_confidence", f64),
                ("combine_low_confidence", f64)]


def group_call_config(cfg) -> snf_group_call_config_t:
    return snf_group_call_config_t(n_samples=len(cfg.snf_input_info), no_qc=int(bool(cfg.no_qc)),
                                   combine_low_confidence_abs=int(cfg.combine_low_confidence_abs),
                                   combine_output_filtered=int(bool(cfg.combine_output_filtered)),
                                   dev_combine_medians=int(bool(getattr(cfg, "dev_combine_medians", False))),
                                   minsvlen_screen=int(cfg.minsvlen_screen),
                                   combine_high_confidence=float(cfg.combine_high_confidence),
                                   combine_low_confidence=float(cfg.combine_low_confidence))
