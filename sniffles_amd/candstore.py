"""Columnar candidate store of the multi-sample merge (`CombineTask.execute`, reference `parallel.py:444-572`).

The reference walks SNF blocks, bins, flush windows, groups and calls object by object.  Here the candidates of ALL tasks of a
merge (contigs, or the parts of `CombineTask.scatter`) become one table the moment they leave the SNF readers:

  1. `collect`  - one C pass over the blocks' `SVCall` lists fills a record per candidate (`abi.GROUP_CAND_DTYPE`), the ALT pool
                  and the BND mate columns, in the reference's visiting order (block, SV type, reader, list order);
  2. `windows`  - a sort by (task, SV type, block, 100-bp bin) and one linear pass give every flush window of every block
                  (`parallel.py:516-534`); the windows of one (task, SV type) form a chain (kept groups seed the next window);
  3. `resolve`  - chains are cut where no candidate can reach an earlier group (`cluster.chain_cuts`, vectorised) and all
                  sub-chains go to the GPU in one `snf_combine_resolve_batch`: the group of every candidate;
  4. `call`     - `snf_combine_call_groups`: per group the running means, the windows it stays active in, the confidence
                  rules, medians / means / exact stdev, the ALT choice and which candidate speaks for its sample
                  (`SVGroup.add_candidate`, `SVGroup.call`, `sv.py:297-481`; the keep rule of `parallel.py:553-556`);
  5. `emit`     - groups leave in the reference's order (flush event, creation order; the rest at the end per SV type), get
                  their `sv_id`s, and one C pass builds the combined `SVCall` objects (ids chained, genotypes of absent samples
                  from the deepest `_COVERAGE` bin the group saw while it was active).

No Python statement runs per candidate or per group.  `CombineTask._execute_many_objects` keeps the object-by-object replay as
the twin the tests compare with (`SNF_COMBINE_OBJECTS=1` selects it)."""
from __future__ import annotations

import os
import threading
import time

import numpy as np

from . import abi, lib, sv
from .soa import SVT


last_timing = {}     # seconds per phase of the last execute_many (measurement only)


def _walk(tasks, samples_snf, config):
    """Blocks of all tasks in visiting order: per block the readers' `read_blocks` results and the `_COVERAGE` dicts."""
    readers = list(samples_snf.items())
    order = [s["internal_id"] for s in config.snf_input_info]
    pos_of = {sid: k for k, (sid, _) in enumerate(readers)}
    blocks, block_cov, block_task = [], [], []
    for ti, t in enumerate(tasks):
        contig = t.contig
        for block_index in t.block_indices:
            per = [snf.read_blocks(contig, block_index) for _, snf in readers]
            blocks.append(per)
            block_cov.append([per[pos_of[s]][0]["_COVERAGE"] if s in pos_of and per[pos_of[s]] is not None else None for s in order])
            block_task.append(ti)
    return readers, order, blocks, block_cov, np.asarray(block_task, np.int64)


_EXT_DTYPE = np.dtype([("blk", np.int64), ("id_start", np.int64), ("ps_start", np.int64), ("alen", np.int64), ("astart", np.int64), ("typ", np.int32), ("mate0", np.int32),
                       ("mate1", np.int32), ("id_len", np.int32), ("ps_len", np.int32), ("ph_hp", np.int8)], align=True)

_MATE_IDS = {}      # mate contig name -> id, process-wide: the ids inside cached column tables stay valid from merge to merge


class ContigColumns:
    """The candidates of ONE reader on ONE contig as columns, built once from the reader's blocks (`_snf_fast.collect` with the
    reader alone: block, SV type, list order) and kept with the reader: a merge - and every later merge over the same readers, or
    the parts of a scattered one - starts from these tables instead of walking `read_blocks` block by block and reading the
    attributes off ~10^5 `SVCall` objects again.  `blk`: block start of every candidate; `cov`: {block start: `_COVERAGE` dict}."""

    def __init__(self, reader, contig, sid, thr, fast):
        starts = sorted(int(b) for b in reader.block_starts(contig))
        per = [[reader.read_blocks(contig, b)] for b in starts]
        self.cov = {b: (p[0][0]["_COVERAGE"] if p[0] is not None else None) for b, p in zip(starts, per)}
        objs, rec, cblk, ctyp, mate, aoff, apool = fast.collect(per, np.asarray([sid], np.int32), tuple(sv.TYPES), int(thr), _MATE_IDS)
        self.objs = objs
        self.rec = np.frombuffer(rec, abi.GROUP_CAND_DTYPE).copy()
        self.blk = np.asarray(starts, np.int64)[np.frombuffer(cblk, np.int32)] if len(objs) else np.zeros(0, np.int64)
        self.typ = np.frombuffer(ctyp, np.int32).astype(np.int64)
        self.mate = np.frombuffer(mate, np.int32).reshape(-1, 2).copy()
        self.aoff = np.frombuffer(aoff, np.int64).copy()
        self.apool = np.frombuffer(apool, np.uint8).copy() if len(apool) else np.zeros(0, np.uint8)
        self.alt_ascii = not bool((self.apool >= 128).any())      # (the pool holds latin-1 bytes: as text they are these bytes only if ASCII)
        self._dense = {}
        # strings a merged record prints per candidate, as one pool: the id (chained per sample) and the phase set of genotypes[0]
        # (vcf.py:40-51 unpack_phase: None / "NULL" print as "."); hp: the haplotype ("1" puts the ALT allele first), -1 = None
        ids = [c.id.encode("utf-8") for c in objs]
        hp, pss = np.full(len(objs), -1, np.int8), []
        for k, c in enumerate(objs):
            g = c.genotypes.get(0) if c.genotypes else None
            h, ps = (None, None)
            if g is not None and g[5] is not None:
                try:
                    h, ps = g[5]
                except TypeError:
                    h, ps = g[5], None
            if h is not None:
                hp[k] = 1 if h == "1" else 2 if h == "2" else 3
            pss.append(None if ps is None or ps == "NULL" else str(ps).encode("utf-8"))
        self.id_len = np.fromiter((len(b) for b in ids), np.int32, len(ids))
        self.id_start = np.concatenate(([0], np.cumsum(self.id_len, dtype=np.int64)))[:-1] if len(ids) else np.zeros(0, np.int64)
        self.ph_hp = hp
        self.ps_len = np.fromiter((-1 if b is None else len(b) for b in pss), np.int32, len(pss))
        base = int(self.id_len.sum())
        self.ps_start = base + (np.concatenate(([0], np.cumsum(np.maximum(self.ps_len, 0), dtype=np.int64)))[:-1] if len(pss) else np.zeros(0, np.int64))
        self.id_pool = np.frombuffer(b"".join(ids) + b"".join(b for b in pss if b is not None) or b"\0", np.uint8)
        # everything but the records as ONE table: a merge concatenates two arrays per reader and contig instead of ten
        self.ext = np.zeros(len(objs), _EXT_DTYPE)
        for name, col in (("blk", self.blk), ("typ", self.typ), ("mate0", self.mate[:, 0]), ("mate1", self.mate[:, 1]), ("id_start", self.id_start),
                          ("id_len", self.id_len), ("ph_hp", self.ph_hp), ("ps_start", self.ps_start), ("ps_len", self.ps_len),
                          ("alen", self.aoff[1:] - self.aoff[:-1]), ("astart", self.aoff[:-1])):
            self.ext[name] = col

    def dense_coverage(self, cb: int):
        """The blocks' `_COVERAGE` dicts as one int32 vector per contig: entry `bin // cb` for every key that is a multiple of `cb`
        (the only keys a merge with this bin size can ask for), -1 elsewhere.  A lookup then costs an index (`group_calls`)."""
        if cb not in self._dense:
            keys = [k for c in self.cov.values() if c for k in c if k % cb == 0 and k >= 0]
            if not keys:
                self._dense[cb] = np.full(1, -1, np.int32)
            else:
                v = np.full(max(keys) // cb + 1, -1, np.int32)
                for c in self.cov.values():
                    if c:
                        ks = np.fromiter((k for k in c if k % cb == 0 and k >= 0), np.int64)
                        v[ks // cb] = np.fromiter((c[k] for k in ks.tolist()), np.int64).astype(np.int32)
                self._dense[cb] = v
        return self._dense[cb]


# How many contigs' column tables a reader keeps (least recently used first out; None: all).  A cached table holds every candidate
# object, record, ALT byte and coverage vector of a contig - the SNF file's content, in memory; the reference streams a contig's blocks
# and lets them go.  The merge driver (`pipeline.combine`) drops the tables when the merge has been written (`clear_columns`); a caller
# that merges the same readers again (bench.py --config 4: candidates resident as columns) keeps them.  `SNF_COLUMNS_CACHE=N` bounds the
# number of contigs a reader holds at a time (0: nothing is kept between calls).
COLUMNS_CACHE_CONTIGS = None
_COLUMNS_LOCK = threading.Lock()


def reader_columns(reader, contig, sid, thr, fast):
    """Cached `ContigColumns` of a reader that can list its blocks (`block_starts(contig)`); None for any other reader.
    The cache lives on the reader (keyed by contig, sample id, support threshold and the identity of the reader's block index, so a
    reader that re-reads its header starts afresh), is bounded (`COLUMNS_CACHE_CONTIGS`) and can be dropped with `clear_columns`.
    The cached candidates are SHARED between the merges that use them: `group_calls` writes the default genotype into
    `genotypes[0]` of candidates that lack one, which every later merge then sees (the value is the same for all of them)."""
    if not hasattr(reader, "block_starts") or getattr(reader, "reqc", False):
        return None
    with _COLUMNS_LOCK:      # (the runs of a merge may be on two threads, SNF_COMBINE_CHUNKS: one builds or evicts at a time)
        return _reader_columns_locked(reader, contig, sid, thr, fast)


def _reader_columns_locked(reader, contig, sid, thr, fast):
    cache = reader.__dict__.setdefault("_snf_columns", {})
    key = (contig, int(sid), int(thr), id(getattr(reader, "index", None)))
    if key in cache:
        cache[key] = cache.pop(key)             # (most recently used last)
        return cache[key]
    cols = ContigColumns(reader, contig, sid, thr, fast)
    keep = os.environ.get("SNF_COLUMNS_CACHE", COLUMNS_CACHE_CONTIGS)
    keep = None if keep is None else int(keep)
    if keep is None or keep > 0:
        cache[key] = cols
        contigs = []
        for k in cache:
            if k[0] not in contigs:
                contigs.append(k[0])
        for old in contigs[:-keep] if keep is not None and len(contigs) > keep else ():
            for k in [k for k in cache if k[0] == old]:
                del cache[k]
    return cols


def clear_columns(reader=None) -> None:
    """Drop the cached column tables of `reader` (every table of the process-wide mate-contig ids stays valid)."""
    if reader is not None:
        reader.__dict__.pop("_snf_columns", None)


def _collect_from_columns(tasks, samples_snf, config, fast):
    """What `_walk` + `collect` return, assembled from the readers' cached tables; None when a reader cannot provide them."""
    readers = list(samples_snf.items())
    order = [s["internal_id"] for s in config.snf_input_info]
    pos_of = {sid: k for k, (sid, _) in enumerate(readers)}
    thr = int(config.combine_support_threshold)
    tabs = {}
    for t in tasks:
        for sid, r in readers:
            if (sid, t.contig) not in tabs:
                c = reader_columns(r, t.contig, sid, thr, fast)
                if c is None:
                    return None
                tabs[(sid, t.contig)] = c
    cb = int(config.coverage_binsize_combine)
    whole = _collect_whole_tables(tasks, readers, order, pos_of, tabs, cb, config)
    if whole is not None:
        return whole
    block_cov, block_task, parts = [], [], []
    eb0 = 0
    dense, eb_start = [], []
    for ti, t in enumerate(tasks):
        bis = np.asarray(t.block_indices, np.int64)
        dense.append([tabs[(s, t.contig)].dense_coverage(cb) if s in pos_of else None for s in order])
        eb_start.append(bis)
        block_task.append(np.full(len(bis), ti, np.int64))
        regular = len(bis) > 1 and bool(np.all(np.diff(bis) == bis[1] - bis[0]))      # the usual range(start, end + bs, bs)
        for sid, _ in readers:
            c = tabs[(sid, t.contig)]
            if not len(c.objs):
                continue
            if regular:
                step = int(bis[1] - bis[0])
                eb = (c.blk - bis[0]) // step
                ok = (c.blk >= bis[0]) & (c.blk <= bis[-1]) & ((c.blk - bis[0]) % step == 0)
            else:
                eb = np.searchsorted(bis, c.blk)
                ok = (eb < len(bis)) & (bis[np.minimum(eb, len(bis) - 1)] == c.blk)      # candidates of the blocks this task holds
            if ok.all():
                parts.append((c, None, eb + eb0))
            else:
                parts.append((c, np.flatnonzero(ok), eb[ok] + eb0))
        eb0 += len(bis)
    block_task = np.concatenate(block_task) if block_task else np.zeros(0, np.int64)
    block_cov = [None] * int(eb0)          # (the dicts themselves are not consulted: covx below)
    covx = dict(dense=dense, eb_task=np.ascontiguousarray(block_task, np.int32), eb_start=np.ascontiguousarray(np.concatenate(eb_start) if eb_start else np.zeros(0), np.int64),
                cb=cb, block_size=int(config.snf_block_size), ids=None)
    if not parts:
        return readers, order, block_cov, block_task, ([], None, None, None, None, None, None), covx
    sizes = [len(c.objs) if idx is None else len(idx) for c, idx, _ in parts]
    offs = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    n = int(offs[-1])
    rec = np.empty(n, abi.GROUP_CAND_DTYPE)
    cblk, ctyp, mate = np.empty(n, np.int32), np.empty(n, np.int32), np.empty((n, 2), np.int32)
    id_start, id_len, ph_hp, ps_start, ps_len = np.empty(n, np.int64), np.empty(n, np.int32), np.empty(n, np.int8), np.empty(n, np.int64), np.empty(n, np.int32)
    lens = np.empty(n, np.int64)
    objs, pools, id_pools = [], [], []
    id_base = 0
    for (c, idx, eb), a, b in zip(parts, offs[:-1].tolist(), offs[1:].tolist()):
        sel = slice(None) if idx is None else idx
        rec[a:b] = c.rec[sel]; cblk[a:b] = eb; ctyp[a:b] = c.typ[sel]; mate[a:b] = c.mate[sel]
        id_start[a:b] = c.id_start[sel] + id_base; id_len[a:b] = c.id_len[sel]; ph_hp[a:b] = c.ph_hp[sel]
        ps_start[a:b] = c.ps_start[sel] + id_base; ps_len[a:b] = c.ps_len[sel]
        id_pools.append(c.id_pool); id_base += len(c.id_pool)
        al = c.aoff[1:] - c.aoff[:-1]
        lens[a:b] = al[sel]
        if idx is None:
            objs.extend(c.objs)
            pools.append(c.apool[:int(c.aoff[-1])])
        else:
            objs.extend([c.objs[i] for i in idx.tolist()])
            st, l = c.aoff[:-1][idx], al[idx]
            tot = int(l.sum())
            pools.append(c.apool[np.repeat(st - (np.cumsum(l) - l), l) + np.arange(tot)] if tot else np.zeros(0, np.uint8))
    aoff = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    apool = np.concatenate(pools) if pools else np.zeros(0, np.uint8)
    id_cols = (np.concatenate(id_pools), id_start, id_len, ph_hp, ps_start, ps_len)
    covx["ids"] = id_cols
    covx["alt_ascii"] = all(c.alt_ascii for c, _, _ in parts)
    return readers, order, block_cov, block_task, (objs, rec, cblk, ctyp, mate, aoff, apool), covx


class _PoolParts:
    """Strings of a merge that still lie in the pools of their tables: pool, first byte and length of every string."""

    def __init__(self, pools, part, start, length):
        self.pools, self.part, self.start, self.length = pools, part, start, length


def _collect_whole_tables(tasks, readers, order, pos_of, tabs, cb, config):
    """`_collect_from_columns` for the usual merge - every task holds a regular run of blocks (`range(start, end + bs, bs)`) that contains
    all candidates of its contig's tables: the tables are concatenated whole (two `np.concatenate` over all readers and contigs) and
    block numbers, id offsets and ALT offsets follow from per-part scalars.  None when a task holds something else (regions, the parts
    of `scatter`): the caller then walks part by part."""
    n_t = len(tasks)
    bis0, step, last, eb0 = np.zeros(n_t, np.int64), np.ones(n_t, np.int64), np.zeros(n_t, np.int64), np.zeros(n_t, np.int64)
    dense, eb_start, block_task = [], [], []
    nb = 0
    for ti, t in enumerate(tasks):
        bis = np.asarray(t.block_indices, np.int64)
        if len(bis) == 0 or (len(bis) > 1 and not bool(np.all(np.diff(bis) == bis[1] - bis[0]))) or (len(bis) > 1 and bis[1] <= bis[0]):
            return None
        bis0[ti], last[ti], eb0[ti] = bis[0], bis[-1], nb
        step[ti] = bis[1] - bis[0] if len(bis) > 1 else 1
        nb += len(bis)
        dense.append([tabs[(s, t.contig)].dense_coverage(cb) if s in pos_of else None for s in order])
        eb_start.append(bis)
        block_task.append(np.full(len(bis), ti, np.int64))
    parts, p_task = [], []
    for ti, t in enumerate(tasks):
        for sid, _ in readers:
            c = tabs[(sid, t.contig)]
            if len(c.objs):
                parts.append(c); p_task.append(ti)
    block_task = np.concatenate(block_task)
    covx = dict(dense=dense, eb_task=np.ascontiguousarray(block_task, np.int32), eb_start=np.ascontiguousarray(np.concatenate(eb_start), np.int64),
                cb=cb, block_size=int(config.snf_block_size), ids=None)
    block_cov = [None] * int(nb)
    if not parts:
        return readers, order, block_cov, block_task, ([], None, None, None, None, None, None), covx
    sizes = np.fromiter((len(c.objs) for c in parts), np.int64, len(parts))
    # (as bytes: numpy copies structured elements field by field - 10 ms for these tables - and bytes with memcpy)
    ext = np.concatenate([c.ext.view(np.uint8) for c in parts]).view(_EXT_DTYPE)
    t_of = np.repeat(np.asarray(p_task, np.int64), sizes)
    d = ext["blk"] - bis0[t_of]
    st = step[t_of]
    if not bool(((d >= 0) & (ext["blk"] <= last[t_of]) & (d % st == 0)).all()):
        return None                                   # a table reaches beyond the blocks of its task
    rec = np.concatenate([c.rec.view(np.uint8) for c in parts]).view(abi.GROUP_CAND_DTYPE)
    cblk = (d // st + eb0[t_of]).astype(np.int32)
    pool_len = np.fromiter((len(c.id_pool) for c in parts), np.int64, len(parts))
    id_base = np.repeat(np.cumsum(pool_len) - pool_len, sizes)
    objs = []
    for c in parts:
        objs.extend(c.objs)
    mate = np.stack((ext["mate0"], ext["mate1"]), axis=1)
    # the ALT bytes stay in the pools of their tables until the sort order is known (`_PoolParts`: gathered once, in that order)
    aoff = None
    apool = _PoolParts([c.apool for c in parts], np.repeat(np.arange(len(parts), dtype=np.int32), sizes), np.ascontiguousarray(ext["astart"]),
                       np.ascontiguousarray(ext["alen"]))
    covx["ids"] = (np.concatenate([c.id_pool for c in parts]), ext["id_start"] + id_base, np.ascontiguousarray(ext["id_len"]),
                   np.ascontiguousarray(ext["ph_hp"]), ext["ps_start"] + id_base, np.ascontiguousarray(ext["ps_len"]))
    covx["alt_ascii"] = all(c.alt_ascii for c in parts)
    return readers, order, block_cov, block_task, (objs, rec, cblk, np.ascontiguousarray(ext["typ"]), mate, aoff, apool), covx


def _regenotype(blocks, readers, config, device):
    """`--reqc`: candidates of SNF files older than 2.5.3 are genotyped again (parallel.py:507-508), one launch."""
    old = [k for k, (_, snf) in enumerate(readers) if getattr(snf, "reqc", False)]
    if not old:
        return
    thr = config.combine_support_threshold
    todo = [c for per in blocks for k in old if per[k] is not None for block in per[k] for t in sv.TYPES for c in block[t]
            if c.support >= thr]
    if todo:
        from . import postprocessing
        postprocessing.genotype_svs(todo, config, device=device)


def _segmented_running(values, seg, take_max: bool):
    """Prefix maximum (or minimum) inside segments; `seg` non-decreasing segment numbers."""
    big = np.int64(1) << 40
    v = values.astype(np.int64)
    if take_max:
        return np.maximum.accumulate(v + seg * big) - seg * big
    return -(np.maximum.accumulate(-v + seg * big) - seg * big)


def execute_many(tasks: list, samples_snf: dict, text_writer=None) -> list:
    """`CombineTask.execute` of every task of `tasks` (one merge: same config and readers); the calls per task.
    The cyclic garbage collector is off for the duration (`sv.no_gc`): the merge creates ~10^6 acyclic containers next to the
    millions the loaded SNF blocks consist of, and every full sweep over those costs as much as the merge itself.
    `text_writer` (a `vcf.VCF` whose `can_write_merged()` holds): no `SVCall` objects are built - per task `(lines: bytes,
    line_off, pos)`, the VCF records `write_call` would print for them, straight from the group table (`VCF.write_merged`)."""
    with sv.no_gc():
        chunks = _task_chunks(tasks, samples_snf)
        if len(chunks) == 1:
            return _execute_many(tasks, samples_snf, text_writer)
        # SNF_COMBINE_CHUNKS=k (an experiment, off by default): tasks share nothing (a chain never leaves its contig task), so the merge
        # of a run of tasks is a merge of its own.  Two threads take the runs in turn - while one waits for the group assignment of its
        # run on the GPU (the C-ABI call releases the interpreter lock; the library admits one such call per device at a time) the
        # other sorts, cuts or formats its own.  It loses: the launch lasts as long as its slowest sub-chain whatever the number of
        # problems, so k runs pay that k times (10 samples x 24 contigs: 156 ms in one launch, 181 / 216 / 261 ms in 2 / 3 / 4 runs)
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        timings = []

        def one(chunk):
            out = _execute_many(chunk, samples_snf, text_writer, timings)
            return out
        with ThreadPoolExecutor(2) as pool:
            parts = list(pool.map(one, chunks))
        last_timing.clear()
        for tm in timings:
            for k, v in tm.items():
                last_timing[k] = last_timing.get(k, 0) + v
        last_timing["chunks"] = len(chunks)
        last_timing["wall"] = time.perf_counter() - t0
        return [r for part in parts for r in part]


def _task_chunks(tasks, samples_snf) -> list:
    """`tasks` as SNF_COMBINE_CHUNKS (default 1) consecutive runs of about equal numbers of blocks."""
    weights = [max(1, len(t.block_indices)) for t in tasks]
    k = max(1, min(int(os.environ.get("SNF_COMBINE_CHUNKS", "1")), len(tasks)))
    if k == 1:
        return [list(tasks)]
    total, out, cur, acc = float(sum(weights)), [], [], 0.0
    for t, w in zip(tasks, weights):
        cur.append(t); acc += w
        if len(out) < k - 1 and acc >= total * (len(out) + 1) / k:
            out.append(cur); cur = []
    if cur:
        out.append(cur)
    return out


def _text_threads(n_records: int) -> int:
    """Threads that format the merged records when they come from arrays alone (`_snf_fast.group_calls`, outside the interpreter
    lock): SNF_TEXT_THREADS, else one per ~2 000 records up to eight and half the cores."""
    env = os.environ.get("SNF_TEXT_THREADS")
    if env is not None:
        return max(1, int(env))
    return max(1, min(8, (os.cpu_count() or 2) // 2, n_records // 2000))


def _record_timing(timings, marks, **counts) -> None:
    """Phases of one `_execute_many`: into `last_timing`, or (a chunk of a merge) appended to `timings` for the caller to add up."""
    d = {name: t1 - t0 for (_, t0), (name, t1) in zip(marks[:-1], marks[1:])}
    d.update(counts)
    if timings is not None:
        timings.append(d)
    else:
        last_timing.clear()
        last_timing.update(d)


def _execute_many(tasks: list, samples_snf: dict, text_writer=None, timings: list = None) -> list:
    fast = sv._load_fast()
    t0 = tasks[0]
    config, device = t0.config, t0.device
    n_tasks = len(tasks)
    tm = [("start", time.perf_counter())]
    mark = lambda name: tm.append((name, time.perf_counter()))  # noqa: E731
    cached = _collect_from_columns(tasks, samples_snf, config, fast) if os.environ.get("SNF_COMBINE_NO_COLUMNS", "0") != "1" else None
    if cached is not None:      # readers that list their blocks: their candidates are resident as columns (ContigColumns)
        readers, order, block_cov, block_task, (objs, rec, cblk, ctyp, mate, aoff, apool), covx = cached
        mark("walk_blocks")
        n = len(objs)
        mark("collect_columns")
        if n == 0:
            return [[] for _ in tasks]
        cblk = cblk.astype(np.int64); ctyp = ctyp.astype(np.int64)
        if not isinstance(apool, _PoolParts):
            aoff, apool = np.ascontiguousarray(aoff, np.int64), np.ascontiguousarray(apool, np.uint8)
    else:
        covx = None
        readers, order, blocks, block_cov, block_task = _walk(tasks, samples_snf, config)
        mark("walk_blocks")
        _regenotype(blocks, readers, config, device)
        sids = np.asarray([sid for sid, _ in readers], np.int32)
        objs, rec, cblk, ctyp, mate, aoff, apool = fast.collect(blocks, sids, tuple(sv.TYPES), int(config.combine_support_threshold), _MATE_IDS)
        n = len(objs)
        mark("collect_columns")
        if n == 0:
            return [[] for _ in tasks]
        rec = np.frombuffer(rec, abi.GROUP_CAND_DTYPE)
        cblk = np.frombuffer(cblk, np.int32).astype(np.int64)
        ctyp = np.frombuffer(ctyp, np.int32).astype(np.int64)
        mate = np.frombuffer(mate, np.int32).reshape(-1, 2)
    ctask = block_task[cblk]
    # ---- 2. sort into chain-major order (task, SV type, block, bin, visiting order) and cut the flush windows
    bin_min = int(config.combine_min_size)
    cbin = np.trunc(rec["pos"] / bin_min).astype(np.int64) * bin_min          # int(pos / bin_min_size) * bin_min_size
    bin_no = cbin // bin_min
    if n and int(ctask.max()) < (1 << 12) and int(cblk.max()) < (1 << 22) and 0 <= int(bin_no.min()) and int(bin_no.max()) < (1 << 26):
        # one key of 63 bits: one stable sort instead of five (a radix sort that skips the digits all keys share)
        perm = np.frombuffer(fast.argsort_i64(np.ascontiguousarray((((ctask * 8 + ctyp) << 22 | cblk) << 26) | bin_no, np.int64)), np.int64)
    else:
        perm = np.lexsort((np.arange(n), cbin, cblk, ctyp, ctask))
    # (np.take: a structured array indexed with [perm] is copied field by field, 20x slower)
    rec, cblk, ctyp, ctask, cbin, mate = np.take(rec, perm), cblk[perm], ctyp[perm], ctask[perm], cbin[perm], np.take(mate, perm, axis=0)
    id_cols = None
    if covx is not None and covx.get("ids") is not None:
        id_cols = (covx["ids"][0],) + tuple(np.ascontiguousarray(a[perm]) for a in covx["ids"][1:])
    # the candidate objects in table order - unless the records are formatted from arrays alone (text, candidate and head columns, no
    # RNAMES): then nothing reads them and the list only stands for its length
    arrays_only = (text_writer is not None and id_cols is not None and bool(covx.get("alt_ascii")) and not getattr(config, "output_rnames", False))
    if arrays_only:
        objs = [None] * n
    else:
        import operator
        objs = list(operator.itemgetter(*perm.tolist())(objs)) if n > 1 else list(objs)
    if isinstance(apool, _PoolParts):
        aoff, apool = fast.gather_pool_parts(apool.pools, apool.part, apool.start, apool.length, np.ascontiguousarray(perm, np.int64))
    else:
        aoff, apool = fast.gather_pool(aoff, apool, np.ascontiguousarray(perm, np.int64))
    aoff = np.frombuffer(aoff, np.int64)
    key = (ctask * 8 + ctyp) * (np.int64(1) << 32) + cblk
    max_cands = max(25, int(len(config.snf_input_info) * 0.5))
    wend, wbin, wsize = fast.flush_windows(np.ascontiguousarray(key), np.ascontiguousarray(cbin, np.int32), bin_min, max_cands,
                                           bool(getattr(config, "combine_exhaustive", False)))
    wend = np.frombuffer(wend, np.int64)
    win_bin = np.frombuffer(wbin, np.int32)
    nw = len(wend)
    win_off = np.concatenate(([0], wend)).astype(np.int64)
    win_thr = np.maximum(np.frombuffer(wsize, np.int32) * 0.5, float(config.combine_overlap_abs))
    cand_win = np.repeat(np.arange(nw, dtype=np.int32), np.diff(win_off))
    w_first = win_off[:-1]
    w_task, w_typ, w_blk = ctask[w_first], ctyp[w_first], cblk[w_first]
    w_chain = w_task * 8 + w_typ                                              # non-decreasing
    chain_first = np.r_[True, w_chain[1:] != w_chain[:-1]]
    chain_no = np.cumsum(chain_first) - 1                                     # chain index per window
    chain_lo = np.flatnonzero(chain_first)
    chain_hi = np.r_[chain_lo[1:], nw]
    n_chains = len(chain_lo)
    # ---- 3. cut the chains where no candidate can reach an earlier group (cluster.chain_cuts) and resolve on the GPU
    sub_first = chain_first.copy()
    if os.environ.get("SNF_COMBINE_NO_CUT", "0") != "1" and nw > 1:
        pos = rec["pos"].astype(np.int64)
        lo = np.minimum.reduceat(pos, w_first)
        hi = np.maximum.reduceat(pos, w_first)
        hi_run = _segmented_running(hi, chain_no, True)                       # prefix maximum inside the chain
        rev = slice(None, None, -1)
        lo_run = _segmented_running(lo[rev], (n_chains - 1 - chain_no)[rev], False)[rev]    # suffix minimum inside the chain
        bnd = w_typ == sv.TYPES.index("BND")
        gate = np.where(bnd, float(config.cluster_merge_bnd) * 2, float(config.combine_match_max))
        gate = np.maximum(gate, 0.0) + 1.0
        cut = np.zeros(nw, bool)
        cut[1:] = (lo_run[1:] - hi_run[:-1]) > gate[1:]
        sub_first |= cut
    s_lo = np.flatnonzero(sub_first)
    s_hi = np.r_[s_lo[1:], nw]
    codes = np.asarray([SVT[t] for t in sv.TYPES], np.int32)[w_typ[s_lo]]
    cols = dict(pos=rec["pos"], svlen=rec["svlen"], support=rec["support"], sample_id=rec["sample"], mate_contig=mate[:, 0],
                mate_ref_start=mate[:, 1])
    keep = []
    mark("sort_and_windows")
    n_ids = int(rec["sample"].max()) + 1
    # (the launch works a sub-chain per wave and lasts as long as its slowest one - a window of kilobase insertions, ~60 ms - whatever the
    # order of the problems: putting the heaviest first changed nothing, 67.6 against 68.0 ms, profiles/r05_merge_runs.log)
    arr, out = abi.combine_chain_problems(codes, win_off[s_lo], win_off[s_hi], s_lo, s_hi, cols, (aoff, apool), win_off, win_bin,
                                          win_thr, n_ids, keep)
    # the call waits for the GPU with the interpreter lock released: what does not need its answer is computed meanwhile (the order
    # in which SVGroup.add_candidate sees the candidates - window by window, support descending (stable) inside a window - the rank of the
    # flush events, the ids of the sub-chains)
    failure = []

    def resolve():
        try:
            lib.combine_resolve_batch(config, arr, device=device)
        except BaseException as e:      # noqa: BLE001 - raised again on the caller's thread
            failure.append(e)
    call = threading.Thread(target=resolve)
    call.start()
    try:
        proc = np.argsort(cand_win.astype(np.int64) << 32 | (np.int64(1) << 31) - rec["support"].astype(np.int64), kind="stable")
        cand_sub = np.repeat(np.arange(len(s_lo)), win_off[s_hi] - win_off[s_lo])
        ev_rank = np.empty(nw, np.int64)
        ev_rank[np.lexsort((np.arange(nw), w_typ, w_blk, w_task))] = np.arange(nw)
    finally:
        call.join()
    if failure:
        raise failure[0]
    mark("resolve_groups_gpu")
    # group numbers: sub-chain -> whole merge (creation order inside a chain is the order of the sub-chains)
    gid_local = out[:n].astype(np.int64)
    created = np.maximum.reduceat(gid_local + 1, win_off[s_lo])      # (a sub-chain's candidates are consecutive; none is empty)
    base = np.cumsum(created) - created
    gid = gid_local + base[cand_sub]
    n_groups = int(created.sum())
    # ---- 4. members in the order SVGroup.add_candidate saw them (`proc`, above)
    member = proc[np.argsort(gid[proc], kind="stable")].astype(np.int32)
    counts = np.bincount(gid, minlength=n_groups)
    group_off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    g_first_win = cand_win[member[group_off[:-1]]].astype(np.int64)
    g_chain = chain_no[g_first_win]
    g_hi = chain_hi[g_chain].astype(np.int32)
    mark("membership")
    gout, chosen, pos_mean = lib.combine_call_groups(config, group_off, member, rec, cand_win, g_hi, win_bin, win_thr, device=device)
    mark("call_groups_gpu")
    # ---- 5. emission order (parallel.py:536-572): a flushed group leaves with its window's event - events run block by block,
    # SV type by SV type, window by window - in creation order; what is still active at the end leaves per SV type
    if getattr(config, "combine_consensus", False) and (gout["emit"] == 1).any():
        raise NotImplementedError("--combine-consensus is broken in the reference (sv.py:382 unpacks 7-tuples into 5)")
    g_task = w_task[g_first_win]
    g_typ = w_typ[g_first_win]
    flushed = gout["flush_win"] >= 0
    rank = np.where(flushed, ev_rank[np.where(flushed, gout["flush_win"], 0)], nw + g_typ)
    em = np.flatnonzero(gout["emit"] == 1)
    em = em[np.lexsort((em, rank[em], g_task[em]))]
    per_task = np.bincount(g_task[em], minlength=n_tasks)
    first_of_task = np.concatenate(([0], np.cumsum(per_task)))[:-1]
    start_id = np.asarray([t.sv_id for t in tasks], np.int64)
    sv_ids = start_id[g_task[em]] + (np.arange(len(em)) - first_of_task[g_task[em]])
    task_ids = np.asarray([t.id for t in tasks], np.int64)[g_task[em]]
    if text_writer is not None and getattr(config, "sort", True):
        # text: the ids are given (emission order) - the records themselves are formatted in the order they are written, per task by
        # position with the emission order among equals (`sorted(calls, key=pos)`, stable): a task's text is then one slice of the buffer.
        # With --no-sort the reference writes a task's calls as they were emitted (result.py:139, :148): that order is kept
        by_pos = np.lexsort((np.arange(len(em)), gout["pos"][em], g_task[em]))
        em, sv_ids, task_ids = em[by_pos], sv_ids[by_pos], task_ids[by_pos]
    # events a group was active in: its first window .. the flush window (or the chain's last); pos_mean at an event = after the
    # last candidate added up to that window; coverage bin of parallel.py:543
    last_win = np.where(flushed, gout["flush_win"], g_hi - 1).astype(np.int64)
    n_ev = (last_win[em] - g_first_win[em] + 1)
    ev_off = np.concatenate(([0], np.cumsum(n_ev))).astype(np.int64)
    ev_group = np.repeat(em, n_ev)
    ev_win = np.repeat(g_first_win[em], n_ev) + (np.arange(int(ev_off[-1])) - np.repeat(ev_off[:-1], n_ev))
    bigw = np.int64(nw + 1)
    m_group = np.repeat(np.arange(n_groups), counts)
    m_key = m_group * bigw + cand_win[member]
    idx = np.searchsorted(m_key, ev_group * bigw + ev_win, side="right") - 1
    cb = int(config.coverage_binsize_combine)
    ev_bin = (np.trunc(pos_mean[idx] / cb).astype(np.int64) * cb).astype(np.int32)
    ev_block = w_blk[ev_win].astype(np.int32)
    sample_ids = np.asarray(order, np.int32)
    spos = np.full(max(int(sample_ids.max(initial=0)), int(rec["sample"].max())) + 2, -1, np.int32)
    spos[sample_ids] = np.arange(len(sample_ids), dtype=np.int32)
    mark("emission_order")
    topt = text_writer.merged_text_options() if text_writer is not None else None
    if topt is not None and id_cols is not None:      # candidate columns: the records are formatted from arrays (no candidate object is read)
        topt.update(rec=np.ascontiguousarray(rec), id_pool=np.ascontiguousarray(id_cols[0]), id_start=id_cols[1], id_len=id_cols[2],
                    ph_hp=id_cols[3], ph_ps_start=id_cols[4], ph_ps_len=id_cols[5])
        if covx.get("alt_ascii"):      # CHROM / ID / ALT of a record from arrays too: no candidate object is read
            topt.update(contigs=[t.contig for t in tasks], types=tuple(sv.TYPES), em_task=np.ascontiguousarray(g_task[em], np.int32),
                        em_typ=np.ascontiguousarray(g_typ[em], np.int32), alt_off=np.ascontiguousarray(aoff, np.int64), alt_pool=apool,
                        threads=_text_threads(len(em)))
    gc_covx = None if covx is None else {k: v for k, v in covx.items() if k not in ("ids", "alt_ascii")}
    calls = fast.group_calls(sv.SVCall, sv.ForwardDifferenceWelford, objs, np.ascontiguousarray(gout), np.ascontiguousarray(em, np.int64),
                             group_off, member, chosen, np.ascontiguousarray(sv_ids, np.int64), np.ascontiguousarray(task_ids, np.int64),
                             sample_ids, spos, block_cov, ev_off, ev_block, np.ascontiguousarray(ev_bin),
                             int(config.combine_null_min_coverage), str(config.id_prefix), len(config.snf_input_info) == 1,
                             np.ascontiguousarray(rec["sample"], np.int32), topt, gc_covx)
    if topt is not None:
        text, line_off, line_pos = calls
        line_off, line_pos = np.frombuffer(line_off, np.int64), np.frombuffer(line_pos, np.int64)
        mark("build_svcalls")
        _record_timing(timings, tm, candidates=n, windows=nw, sub_chains=len(s_lo), groups=n_groups, calls=len(em))
        result = []
        for k, t in enumerate(tasks):
            a, b = int(first_of_task[k]), int(first_of_task[k] + per_task[k])
            t.sv_id += int(per_task[k])
            result.append((text, line_off[a:b + 1], line_pos[a:b]))
        return result
    if config.combine_pair_relabel:
        thr = config.combine_pair_relabel_threshold
        for call in calls:                                                     # sv.py:416-426 (an option; off by default)
            top = (0, 0)
            for a, b, q, *_ in call.genotypes.values():
                if q > thr and a != ".":
                    top = max(top, (a, b))
            if top != (0, 0):
                for sid, (a, b, q, dr, dv, ps, nid) in list(call.genotypes.items()):
                    if q < thr and a != ".":
                        call.genotypes[sid] = (top[0], top[1], q, dr, dv, ps, nid)
    mark("build_svcalls")
    _record_timing(timings, tm, candidates=n, windows=nw, sub_chains=len(s_lo), groups=n_groups, calls=len(calls))
    result = []
    for k, t in enumerate(tasks):
        a, b = int(first_of_task[k]), int(first_of_task[k] + per_task[k])
        t.sv_id += int(per_task[k])
        result.append(calls[a:b])
    return result
