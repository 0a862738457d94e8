"""bench.py --config 4: population merge (BASELINE.json configs[4]): S HG002-shaped samples -> combined multi-sample calls.

Set-up (untimed): every sample of the population (same SV sites and alleles, own reads; 70 % of the sites per sample) runs
through the calling hot path on this GPU; its candidates go into 100-kb SNF blocks with the downsampled coverage
(`snf_batch_block_coverage`) exactly as `CallTask.write_snf_part` stores them - kept in memory behind the SNF reader interface
instead of gzip / pickle files (container I/O is not what is measured).
One step = `CombineTask.execute` of every contig task of this rank over the S readers (`parallel.py:444-572`): the block / bin /
flush-window walk and the `SVGroup.call` replay on the host, the group assignment with its on-demand banded edit distances in
ONE `snf_combine_resolve_batch` call for all contig tasks of the rank.  N > 1: contigs sharded longest-first over the ranks (weak scaling
would need
N populations; the merge of one population is what `configs[4]` names, so this is STRONG scaling: `scaling: "strong"`).
value = candidates merged per second (whole job); the kernel inside the C-ABI call is reported against the roofline with the
bytes it aligned, DP cells per second next to it (the bound of this kernel is VALU issue, not HBM: integer bit-vector work).
"""
from __future__ import annotations

import os
import sys
import time

from tools.bench_common import DEV, dev_sync

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0


class MemReader:
    """The SNF reader interface `CombineTask.execute` uses (`SNFile.read_blocks`, attribute `reqc`) over blocks in memory."""
    reqc = False

    def __init__(self):
        self.blocks = {}            # contig -> {block_index: block dict}

    def read_blocks(self, contig, block_index):
        b = self.blocks.get(contig, {}).get(block_index)
        return None if b is None else [b]

    def block_starts(self, contig):
        return list(self.blocks.get(contig, {}))


def build_sample(cfg, tasks, device, sid):
    """Candidates of one sample (all contig tasks in one device batch) -> SNF blocks in memory (snf.py:90-98, 249-267)."""
    from sniffles_amd import lib, sv
    reader = MemReader()
    bs, binsize = cfg.snf_block_size, cfg.coverage_binsize_combine
    per_block = bs // binsize
    n = 0
    with lib.Batch(cfg, tasks, device=device) as b:
        b.call_candidates(); b.finalize()
        res = b.fetch(1)
        for t, ti in enumerate(tasks):
            lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
            cands = sv.materialize_candidates(res, ti, lo, hi)
            sv.apply_final(cands, res, ti, lo)
            blocks = reader.blocks.setdefault(ti.contig, {})
            for c in cands:
                c.finalize()
                c.rnames = None
                bi = int(c.pos / bs) * bs
                if bi not in blocks:          # (SNFile.store, snf.py:91-100: a single-break candidate opens its block too - the block
                    blocks[bi] = {svtype: [] for svtype in sv.TYPES}      # then carries the sample's coverage for the other samples' calls)
                    blocks[bi]["_COVERAGE"] = {}
                if c.svtype not in sv.TYPES:
                    continue
                blocks[bi][c.svtype].append(c)
                n += 1
            if blocks:
                first, last = min(blocks) // bs, max(blocks) // bs
                depth = b.block_coverage(t, binsize, first * per_block, (last - first + 1) * per_block)
                for bi, blk in blocks.items():
                    base = (bi // bs - first) * per_block
                    for i in range(per_block):
                        d = int(depth[base + i])
                        if d >= 0:
                            blk["_COVERAGE"][bi + i * binsize] = d
    return reader, n


def run(ctx):
    import torch
    import torch.distributed as dist

    from sniffles_amd import candstore, cluster, dist as sdist, lib, parallel, synth
    from sniffles_amd.config import SnifflesConfig

    args, rank, world, local_rank, use_dist = (ctx[k] for k in ("args", "rank", "world", "local_rank", "use_dist"))
    S = max(2, args.samples)
    cov = args.coverage if args.coverage is not None else 15.0
    steps = args.steps if args.steps is not None else 3
    warmup = args.warmup if args.warmup is not None else 1
    contigs = [(ci, c, max(200000, int(synth.GRCH38[c] * args.scale))) for ci, c in enumerate(synth.CONTIGS)]
    mine = sdist.shard_lpt([L for _, _, L in contigs], world)[rank]
    my_contigs = [contigs[i] for i in mine]
    call_cfg = SnifflesConfig()
    t0 = time.time()
    readers, n_cands = {}, 0
    for s in range(S):
        tasks = [synth.gen_task(ci, c, L, cov, seed=100 + s, site_seed=501) for ci, c, L in my_contigs]
        readers[s], n = build_sample(call_cfg, tasks, local_rank, s)
        n_cands += n
    t_setup = time.time() - t0
    # the loaded SNF blocks are the resident input: take them out of the cyclic collector's sweeps, as a long-running merge
    # process would (gc.freeze); the merge itself runs with the collector off (sniffles_amd.candstore.execute_many)
    import gc
    gc.collect()
    gc.freeze()
    cfg = SnifflesConfig()
    cfg.mode = "combine"
    cfg.snf_input_info = [dict(internal_id=s, sample_id=f"S{s}") for s in range(S)]
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(S)]

    box = dict(calls=0, kernel_ms=0.0, stats=[0, 0, 0, 0], abi_s=0.0)
    real_call = lib.combine_resolve_batch

    def timed_call(config, problems, device=0):     # the C-ABI call alone (host staging + H2D + kernel + D2H)
        t = time.perf_counter()
        real_call(config, problems, device=device)
        box["abi_s"] += time.perf_counter() - t
        st = lib.combine_last_stats(device)
        box["kernel_ms"] += st["kernel_ms"]
        for k, name in enumerate(("alignments", "aligned_bytes", "dp_cells", "staged_bytes")):
            box["stats"][k] += st[name]
    lib.combine_resolve_batch = timed_call

    import io
    from sniffles_amd import vcf
    text_box = [None]

    def one_pass(objects=False):
        """The merge of this rank's contig tasks -> the merged VCF records (what the reference's combine run produces): formatted
        straight from the group table (vcf.VCF.write_merged); objects=True builds the SVCall objects and lets write_call print them."""
        box.update(calls=0, kernel_ms=0.0, stats=[0, 0, 0, 0], abi_s=0.0)
        tasks = [parallel.CombineTask(id=ci, sv_id=0, contig=c, start=0, end=L - 1, config=cfg, device=local_rank) for ci, c,
                 L in my_contigs]
        # the output handle: a text file over a binary buffer, as `open(path, "w")` gives the writer (the merged records are bytes already)
        buf = io.TextIOWrapper(io.BytesIO(), encoding="utf-8", newline="", write_through=True)
        w = vcf.VCF(cfg, buf)
        # the contig tasks of this rank share one group-assignment launch (CombineTask.execute_many)
        if objects or not w.can_write_merged():
            n = 0
            for calls in parallel.CombineTask.execute_many(tasks, readers):
                for c in sorted(calls, key=lambda c: c.pos):
                    n += w.write_call(c)
            box["calls"] = n
        else:
            box["calls"] = sum(w.write_merged(part) for part in parallel.CombineTask.execute_many(tasks, readers, text_writer=w))
        buf.flush()
        text_box[0] = buf                         # (the records are in the handle when the pass ends; read back outside the timed passes)

    def barrier():
        if use_dist:
            dist.barrier()
        dev_sync(torch)

    for _ in range(warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    barrier()
    dt = time.perf_counter() - t0
    phases = dict(candstore.last_timing)
    text_fast = text_box[0].buffer.getvalue()
    t1 = time.perf_counter()
    one_pass(objects=True)                 # for the record (and as a check): the same records through SVCall objects + write_call
    objects_ms = (time.perf_counter() - t1) * 1e3
    text_equal = text_box[0].buffer.getvalue() == text_fast
    lib.combine_resolve_batch = real_call
    tt = torch.tensor([dt], dtype=torch.float64, device=ctx.get("ctl_dev", DEV))
    tot = torch.tensor([n_cands, box["calls"]], dtype=torch.int64, device=ctx.get("ctl_dev", DEV))
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    dt = float(tt.item())
    total_cands, total_calls = int(tot[0].item()), int(tot[1].item())
    kms = box["kernel_ms"]
    al, ab, cells, staged = box["stats"]
    achieved = ab / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    out = dict(metric="SV candidates merged/sec (multi-sample combine: candidates resident as columns -> group assignment with banded "
                      "edit distance -> SVGroup.call -> merged VCF records)",
               value=total_cands * steps / dt, unit="candidates/s", n_gpus=world, steps=steps, warmup=warmup,
               ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="int32/f64/u64 bit-vectors",
               data="synthetic",
               config=dict(workload=f"population merge: {S} HG002-shaped samples at {cov:g}x (shared SV sites, own reads) -> combine "
                                    f"(BASELINE.json configs[4]), candidates from this package's calling path, SNF blocks in memory",
                           python_gc="SNF blocks frozen after loading (gc.freeze); collector off inside execute_many",
                           baseline_config=4, samples=S, coverage=cov, scale=args.scale, contig_tasks=len(contigs),
                           candidates=total_cands, combined_calls=total_calls, setup_s=round(t_setup, 1),
                           parallelism=f"contig tasks sharded longest-first over {world} ranks, no data-path collective",
                           host_phases_ms={k: (round(v * 1e3, 1) if isinstance(v, float) else v) for k, v in phases.items()},
                           output="merged VCF records as text straight from the group table (vcf.VCF.write_merged), sorted by position "
                                  "per contig task",
                           through_svcall_objects_ms=round(objects_ms, 1), text_equals_object_path=bool(text_equal),
                           vcf_bytes=len(text_fast or ""),
                           rank0=dict(c_abi_call_ms=round(box["abi_s"] * 1e3, 2), kernel_ms=round(kms, 3),
                                      host_ms=round(dt / steps * 1e3 - box["abi_s"] * 1e3, 1), staged_bytes=staged,
                                      alignments=al, dp_cells=cells,
                                      dp_cells_per_s=round(cells / (kms * 1e-3)) if kms > 0 else None),
                           parity_unpinned=["edit distance vs edlib itself (edlib absent; pinned to the exact Levenshtein DP)"]),
               roofline=dict(bound="hbm", kernel="combine_problem_wave", achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(achieved / HBM_PEAK_GBS, 6), traffic=None, kernel_ms=round(kms, 3), algorithmic_bytes=ab,
                             note="bit-parallel Myers blocks: the kernel is VALU-issue bound (integer work, ~25 ops per 64 DP cells), "
                                  "not HBM bound; bytes = lengths of the aligned strings"))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], ver = cpu_baseline_and_verify(cfg, readers, my_contigs, S, local_rank, args)
        if ver is not None:
            out["verified"] = bool(ver["ok"] and text_equal)      # group assignment vs the oracle AND the text path vs the object path
            out["verify"] = ver
        if not getattr(args, "no_reference_baseline", False):
            # the UNMODIFIED reference's CombineTask.execute on the same population, on this box's cores (edlib -> the bit-parallel
            # stand-in)
            try:
                ref_base = reference_baseline(my_contigs, S, cov, total_cands, int(phases.get("calls", total_calls)), dt / steps,
                                              sample=getattr(args, "reference_sample_contigs", None), text=text_fast)
            except Exception as e:  # noqa: BLE001 - a baseline that cannot run must not take the line down; it says why
                ref_base = None
                out["cpu_baseline"]["reference_error"] = f"{type(e).__name__}: {str(e)[:400]}"
            if ref_base is not None:
                ref_base["port"] = out["cpu_baseline"]
                out["cpu_baseline"] = ref_base
                vr = ref_base.pop("verified_vs_reference", None)
                if vr is not None:
                    out["verified_vs_reference"] = vr
        # (the driver's record keeps `config` whole and only the names of other keys: the checks' outcome goes there too)
        out["config"]["verified"] = out.get("verified")
        out["config"]["verified_vs_reference"] = (out.get("verified_vs_reference") or {}).get("ok")
    return out


def records_by_contig(text) -> dict:
    """VCF record lines grouped by CHROM, in the order they were written."""
    out = {}
    if isinstance(text, str):
        text = text.encode("utf-8")
    for line in text.split(b"\n"):
        if line and not line.startswith(b"#"):
            out.setdefault(line[:line.index(b"\t")].decode(), []).append(line)
    return out


VCF_COLUMNS = ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT")


def reference_differences(items, contigs, text, n_samples) -> dict:
    """The merged VCF records this package wrote (`text`) against the UNMODIFIED reference's: its `CombineTask.execute`
    (parallel.py:444-572, sv.py:320-481) on the same population, printed by its own writer (vcf.py:216-350) - line by line, every
    column (CHROM POS ID REF ALT QUAL FILTER, INFO incl. SVTYPE / SVLEN / END / SUPPORT / COVERAGE / STDEV_* / AF, the per-sample
    GT:GQ:DR:DV:ID columns with the chained candidate ids)."""
    mine = records_by_contig(text)
    diffs, n_cmp = [], 0
    for ci, c, _ in contigs:
        if ci not in items:
            continue
        exp = records_by_contig(items[ci]["vcf"]).get(c, [])
        got = mine.get(c, [])
        n_cmp += len(exp)
        if len(exp) != len(got):
            diffs.append(f"{c}: {len(exp)} reference records, {len(got)} here")
            continue
        for k, (a, b) in enumerate(zip(got, exp)):
            if a != b:
                fa, fb = a.split(b"\t"), b.split(b"\t")
                cols = [(VCF_COLUMNS[i] if i < len(VCF_COLUMNS) else f"sample {i - len(VCF_COLUMNS)}") for i in range(min(len(fa), len(fb)))
                        if fa[i] != fb[i]]
                first = next((i for i in range(min(len(fa), len(fb))) if fa[i] != fb[i]), None)
                shown = "" if first is None else f" [here {fa[first][:80].decode()!r}, reference {fb[first][:80].decode()!r}]"
                diffs.append(f"{c} record {k} ({fb[2].decode()} at {fb[1].decode()}): {', '.join(cols) or 'column count'}{shown}")
                if len(diffs) > 5:
                    break
    return dict(ok=not diffs, records_compared=n_cmp, contigs_compared=sum(1 for ci, _, _ in contigs if ci in items), differences=diffs[:5],
                what="every merged VCF record (all columns, per-sample genotype columns and id chains included) vs the text the unmodified "
                     "reference's own writer prints for its CombineTask.execute on the same population, on this box")


def reference_baseline(contigs, S, cov, n_cands, n_calls, gpu_s_per_merge, sample=None, text=None):
    """`cpu_baseline` of kind "reference (edlib stand-in)": oracle/ref_combine_pool.py - the unmodified reference's `CombineTask.execute`
    (parallel.py:444-572), one process per contig, on the same seeded population (its samples called by the reference's own path, untimed),
    `sv.align` = the bit-parallel algorithm edlib implements (edlib is absent from this image: parity unpinned)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_combine_pool
    if not ref_combine_pool.available():
        return None
    # a bounded sample (bench.py's default line): the `sample` smallest contig tasks, one process each; the whole workload: bench.py
    # --config 4
    if sample:
        part = sorted(contigs, key=lambda c: c[2])[:int(sample)]
        r = ref_combine_pool.run(part, S, cov, want_records=text is not None)
        ver = reference_differences(r["items"], part, text, S) if text is not None else None
        per_cand = r["hot_single_core_s"] / max(1, r["candidates"])
        share = max(c[2] for c in contigs) / float(sum(c[2] for c in contigs))
        return dict(value=r["candidates"] / r["hot_all_core_s"], unit="candidates/s", cores=r["procs"],
                    kind="reference (edlib stand-in)", host_cores=r["cores"], verified_vs_reference=ver,
                    hot_all_core_s=round(r["hot_all_core_s"], 3), candidates=r["candidates"], combined_calls=r["combined"],
                    single_core_cand_s=round(1.0 / per_cand, 1),
                    # the reference merges one contig per process: the wall clock of a whole merge is its largest contig task
                    whole_merge_estimate=dict(all_core_s=round(per_cand * n_cands * share, 1),
                                              vs_this_package=round(per_cand * n_cands * share / gpu_s_per_merge, 1),
                                              note="single-core seconds per candidate of the sample x the candidates of the largest "
                                                   "contig task (one process "
                                                   "per contig); measured on the whole workload by `bench.py --config 4`: "
                                                   "profiles/r06_final_bench_config4.json"),
                    parity_unpinned="sv.align is oracle/snf_oracle.c::snf_oracle_edit_distance_myers (the algorithm edlib implements, "
                                    "pinned to the exact DP), not edlib",
                    sample=f"the {len(part)} smallest contig tasks ({', '.join(c[1] for c in part)}) x {S} samples, the unmodified "
                           f"reference's CombineTask.execute, one process "
                           f"per contig: slowest process {r['hot_all_core_s']:.2f} s, sum {r['hot_single_core_s']:.1f} s; whole leg "
                           f"{r['total_wall_s']:.0f} s")
    r = ref_combine_pool.run(list(contigs), S, cov, max_procs=int(os.environ.get("SNF_BENCH_REF_PROCS", "0")) or None,
                             want_records=text is not None)
    ver = reference_differences(r["items"], list(contigs), text, S) if text is not None else None
    return dict(value=r["candidates"] / r["hot_all_core_s"], unit="candidates/s", cores=r["procs"], kind="reference (edlib stand-in)",
                host_cores=r["cores"], verified_vs_reference=ver,
                hot_all_core_s=round(r["hot_all_core_s"], 3), hot_single_core_s=round(r["hot_single_core_s"], 2),
                candidates=r["candidates"], combined_calls=r["combined"],
                same_population=dict(candidates_equal=bool(r["candidates"] == n_cands), combined_calls_equal=bool(r["combined"] == n_calls),
                                     here=dict(candidates=n_cands, combined_calls=n_calls)),
                vs_baseline=dict(merge=round(r["hot_all_core_s"] / gpu_s_per_merge, 1),
                                 note="reference all-core seconds for the merge / this package's seconds per merge (candidates resident "
                                      "as columns -> merged VCF records)"),
                parity_unpinned="sv.align is oracle/snf_oracle.c::snf_oracle_edit_distance_myers (the algorithm edlib implements, pinned "
                                "to the exact DP), not edlib",
                sample=f"the whole workload: {len(contigs)} contig tasks x {S} samples, the unmodified reference's CombineTask.execute, "
                       f"one process per contig "
                       f"({r['procs']} processes on {r['cores']} usable cores), every process starts its first merge at a barrier once "
                       f"its samples' SNF blocks "
                       f"exist (built by the reference's own calling path, untimed): slowest process {r['hot_all_core_s']:.2f} s, sum "
                       f"{r['hot_single_core_s']:.1f} s; "
                       f"whole leg {r['total_wall_s']:.0f} s")


def sample_windows(cfg, readers, contigs, limit):
    """Flush windows of the merge as independent resolve_block_groups problems (no kept groups): the binning walk of
    CombineTask.execute over the first contigs, INS windows first (they carry the alignments)."""
    from sniffles_amd import sv
    out = []
    bin_min = cfg.combine_min_size
    cap = max(25, int(len(cfg.snf_input_info) * 0.5))
    for ci, contig, L in contigs:
        blocks = sorted(set(b for r in readers.values() for b in r.blocks.get(contig, {})))
        for bi in blocks:
            for svtype in sv.TYPES:
                bins = {}
                for sid, r in readers.items():
                    blk = r.blocks.get(contig, {}).get(bi)
                    if blk is None:
                        continue
                    for c in blk[svtype]:
                        if c.support >= cfg.combine_support_threshold:
                            c.sample_internal_id = sid
                            bins.setdefault(int(c.pos / bin_min) * bin_min, []).append(c)
                cur = []
                keys = sorted(bins)
                for k in keys:
                    cur.extend(bins[k])
                    if len(cur) >= cap or k == keys[-1]:
                        out.append((svtype, cur))
                        cur = []
                if len(out) >= 4 * limit:
                    break
        if len(out) >= 4 * limit:
            break
    out.sort(key=lambda w: (w[0] != "INS",))
    return out[:limit]


def cpu_baseline_and_verify(cfg, readers, contigs, S, device, args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import combine_pool
    from sniffles_amd import cluster
    cores = os.cpu_count() or 1
    wins = sample_windows(cfg, readers, contigs, limit=max(64, min(2000, 12 * cores)))
    if not wins:
        return dict(value=0.0, unit="candidates/s", cores=0, kind="port", sample="no windows"), None
    n_c = sum(len(c) for _, c in wins)
    r = combine_pool.run_windows([(k, combine_pool.window_of(t, c)) for k, (t, c) in enumerate(wins)], {}, S)
    t0 = time.perf_counter()
    got = cluster.resolve_chains_batch([(t, c, [0, len(c)], [0], [-1.0]) for t, c in wins], cfg, device=device, cut=False)
    t_gpu = time.perf_counter() - t0
    bad = [k for k, ((t, c), g) in enumerate(zip(wins, got)) if list(g[:len(c)]) != r["groups"][k]]
    base = dict(value=n_c / r["slowest_s"], unit="candidates/s", cores=r["procs"], kind="port", cores_used=r["procs"],
                host_cores=r["cores"],
                all_core_cand_s=n_c / r["slowest_s"], single_core_cand_s=n_c / r["sum_s"],
                sample=f"{len(wins)} flush windows ({n_c} candidates, INS windows first) of the same merge as independent "
                       f"resolve_block_groups problems through the C oracle (exact edit-distance DP), {r['procs']} processes: slowest "
                       f"{r['slowest_s']:.2f} s, sum {r['sum_s']:.2f} s; the same windows on the GPU incl. packing: {t_gpu:.3f} s")
    ver = dict(ok=not bad, windows_compared=len(wins), candidates_compared=n_c,
               what="group assignment of every candidate of the sampled windows, GPU vs the C oracle", differences=bad[:5])
    return base, ver
