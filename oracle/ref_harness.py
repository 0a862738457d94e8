"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

Runs the UNMODIFIED reference implementation (`/root/reference/src/sniffles`)
on a `sniffles_amd.soa.TaskInput` and returns canonical result records.  Usable
where the reference checkout or its staged build (oracle/_ref, see make_ref.py) is present; it is how
`oracle/make_golden.py` produces the committed fixtures under `tests/golden/`
that pin the C restatement (`oracle/snf_oracle.c`) and, through it, the HIP
path.

pysam / spoa / edlib are not installed: pysam and spoa are stubbed exactly as
SURVEY.md Appendix C describes (10 CIGAR constants + 4 empty classes; a `poa`
symbol); edlib is replaced by an exact Levenshtein DP (edlib's default
mode="NW", task="distance" is global unit-cost edit distance).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import make_ref  # noqa: E402

# the reference checkout in the build container; elsewhere the byte-compiled, unmodified copy `oracle/make_ref.py` staged
# under the git-ignored oracle/_ref/ (it travels to the GPU box like a built .so)
REF_SRC = make_ref.ref_root() or make_ref.SRC_ROOT


def reference_available() -> bool:
    return make_ref.ref_root() is not None


def reference_kind() -> str:
    """"checkout" (build container), "staged" (oracle/_ref, compiled by make_ref.py) or "absent"."""
    r = make_ref.ref_root()
    return "absent" if r is None else ("checkout" if r == make_ref.SRC_ROOT else "staged")


_loaded = None


def load_reference():
    """Import the reference modules under the stubs; returns a namespace of modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference sources not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    ps = types.ModuleType("pysam")
    for i, n in enumerate("CMATCH CINS CDEL CREF_SKIP CSOFT_CLIP CHARD_CLIP CPAD CEQUAL CDIFF CBACK".split()):
        setattr(ps, n, i)
    for n in "AlignedSegment AlignmentFile FastaFile VariantFile".split():
        setattr(ps, n, type(n, (), {}))
    sys.modules.setdefault("pysam", ps)
    sp = types.ModuleType("spoa")
    sp.poa = lambda *a, **k: None
    sys.modules.setdefault("spoa", sp)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    from sniffles import config, leadprov, cluster, sv, parallel, postprocessing, consensus, genotyping, util
    ns = types.SimpleNamespace(config=config, leadprov=leadprov, cluster=cluster, sv=sv, parallel=parallel,
                               postprocessing=postprocessing, consensus=consensus, genotyping=genotyping, util=util)
    _loaded = ns
    return ns


def levenshtein(a: str, b: str) -> int:
    """Exact global unit-cost edit distance (what edlib.align(a,b)['editDistance'] returns by default)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def make_config(extra_args=(), qc_nm_threshold=0.02):
    ref = load_reference()
    cfg = ref.config.SnifflesConfig("--input", "x.bam", "--vcf", "out.vcf", *extra_args)
    cfg.mode = "call_sample"
    cfg.average_regional_nm = qc_nm_threshold
    cfg.qc_nm_threshold = qc_nm_threshold
    return cfg


def build_leads(ti):
    """TaskInput rows -> reference Lead objects, in arrival order."""
    from sniffles_amd.soa import SVTYPES, SOURCES, SVLEN_NONE
    ref = load_reference()
    Lead, BND = ref.leadprov.Lead, ref.sv.SVCallBNDInfo
    L = ti.leads
    pool = ti.seq_pool.tobytes()
    out = []
    for i in range(ti.n_leads):
        svt = SVTYPES[L["svtype"][i]]
        sl = int(L["seq_len"][i])
        so = int(L["seq_off"][i])
        svlen = int(L["svlen"][i])
        ld = Lead(read_id=int(L["read_id"][i]), read_qname=ti.qname(int(L["qname_id"][i])), contig=ti.contig,
                  ref_start=int(L["ref_start"][i]), ref_end=int(L["ref_end"][i]),
                  qry_start=int(L["qry_start"][i]), qry_end=int(L["qry_end"][i]),
                  strand="-" if L["strand"][i] else "+", mapq=int(L["mapq"][i]), nm=float(L["nm"][i]),
                  source=SOURCES[L["source"][i]], svtype=svt,
                  svlen=None if svlen == int(SVLEN_NONE) else svlen,
                  seq=None if sl < 0 else pool[so:so + sl].decode("latin-1"),
                  hap=str(int(L["hap"][i])), phase_set=ti.ps_name(int(L["ps_rank"][i])),
                  is_sa=bool(L["is_sa"][i]), read_len=int(L["read_len"][i]))
        if svt == "BND":
            ld.bnd_info = BND(mate_contig=ti.contig_name(int(L["mate_contig"][i])),
                              mate_ref_start=int(L["mate_ref_start"][i]),
                              is_first=bool(L["bnd_is_first"][i]), is_reverse=bool(L["bnd_is_reverse"][i]))
        out.append(ld)
    return out


def build_task(ti, cfg):
    """Reference CallTask with a populated LeadProvider (SURVEY.md Appendix C recipe)."""
    ref = load_reference()
    lp = ref.leadprov.LeadProvider(cfg, 0, ti.contig)
    lp.coverage = np.zeros(ti.contig_len, dtype=np.uint16)
    lp.start, lp.end = 0, ti.contig_len
    for s, e in zip(ti.read_start.tolist(), ti.read_end.tolist()):
        lp.coverage[s:e] += 1
    if getattr(ti, "nmask_start", None) is not None and len(ti.nmask_start):
        # the reference's own _mask_N_coverage (leadprov.py:420-443) over a FASTA stand-in that has 'N' exactly on the task's
        # mask intervals (build_leadtab calls it after the regions have been read)
        seq = np.full(ti.contig_len, ord("A"), np.uint8)
        for a, e in zip(ti.nmask_start.tolist(), ti.nmask_end.tolist()):
            seq[a:e] = ord("N")

        class _Fasta:
            def __init__(self, path):
                pass

            def fetch(self, contig, start=None, end=None):
                return seq[start:end].tobytes().decode("ascii")
        keep_ref, keep_cls = cfg.reference, ref.leadprov.pysam.FastaFile
        cfg.reference, ref.leadprov.pysam.FastaFile = "reference.fa", _Fasta
        try:
            lp._mask_N_coverage()
        finally:
            cfg.reference, ref.leadprov.pysam.FastaFile = keep_ref, keep_cls
    bs = cfg.cluster_binsize
    for ld in build_leads(ti):
        # build_leadtab keeps only leads inside the task region (leadprov.py:464-468)
        if lp.start <= ld.ref_start < lp.end:
            lp.record_lead(ld, int(ld.ref_start / bs) * bs)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_hap_ref(hp, int(s / bs) * bs, int(e / bs) * bs, bs)
    task = ref.parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0,
                                 end=ti.contig_len, config=cfg)
    task.lead_provider = lp
    if ti.tr_start is not None:
        task.tandem_repeats = list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    else:
        task.tandem_repeats = None
    return task


def _f(x):
    return None if x is None else float(x)


def call_record(c, stage: str) -> dict:
    """Canonical, JSON-able record of one reference SVCall."""
    info = c.info
    gt = c.genotypes.get(0) if c.genotypes else None
    rec = dict(
        id=c.id, contig=c.contig, pos=int(c.pos), end=int(c.end), svtype=c.svtype, svlen=int(c.svlen),
        support=int(c.support), qual=int(c.qual), precise=bool(c.precise), fwd=int(c.fwd), rev=int(c.rev),
        filter=c.filter, qc=bool(c.qc), nm=_f(c.nm), alt=c.alt,
        stdev_pos=_f(info.get("STDEV_POS")), stdev_len=_f(info.get("STDEV_LEN")),
        support_long=info.get("SUPPORT_LONG"), support_sa=info.get("SUPPORT_SA"),
        cov=[int(c.coverage_upstream), int(c.coverage_start), int(c.coverage_center), int(c.coverage_end),
             int(c.coverage_downstream)],
        rnames=sorted(c.rnames) if c.rnames is not None else None,
        bnd=None if c.bnd_info is None else [c.bnd_info.mate_contig, int(c.bnd_info.mate_ref_start),
                                            bool(c.bnd_info.is_first), bool(c.bnd_info.is_reverse)],
    )
    if stage == "final":
        rec["gt"] = None if gt is None else [gt[0], gt[1], int(gt[2]), int(gt[3]), int(gt[4]),
                                             list(gt[5]) if gt[5] is not None else None]
        rec["vaf"] = _f(info.get("VAF"))
        rec["phase"] = info.get("PHASE")
    return rec


def run_reference(ti, extra_args=(), keep_qc_fails=True, finalize_keep=False, cfg=None):
    """Run call_candidates + finalize_candidates of the reference on one task.

    Returns dict(candidates=[...], final=[...], coverage_average_total=float) or dict(error=str)
    when the reference itself raises (e.g. UnboundLocalError for a BND-first task).
    """
    cfg = cfg or make_config(extra_args, ti.qc_nm_threshold)
    task = build_task(ti, cfg)
    try:
        cands = task.call_candidates(keep_qc_fails, cfg)
    except Exception as e:  # the reference's own failure mode is part of its behaviour
        return dict(error=type(e).__name__)
    out = dict(candidates=[call_record(c, "cand") for c in cands],
               coverage_average_total=float(task.coverage_average_total))
    final = task.finalize_candidates(cands, finalize_keep, cfg)
    out["final"] = [call_record(c, "final") for c in final]
    return out


def run_reference_regenotype(ti, extra_args=()):
    """What --reqc does to the candidates of an old SNF file (parallel.py:507-508): the reference's own
    postprocessing.genotype_sv(cand, config) on every finalized candidate of a task, a second time."""
    ref = load_reference()
    cfg = make_config(tuple(extra_args), ti.qc_nm_threshold)
    task = build_task(ti, cfg)
    cands = task.call_candidates(False, cfg)
    task.finalize_candidates(cands, True, cfg)
    out = []
    for c in cands:
        if c.svtype not in ref.sv.TYPES:
            continue
        ref.postprocessing.genotype_sv(c, cfg)
        gt = c.genotypes.get(0)
        out.append(dict(id=c.id, filter=c.filter, qc=bool(c.qc), vaf=_f(c.info.get("VAF")), phase=c.info.get("PHASE"),
                        gt=None if gt is None else [gt[0], gt[1], int(gt[2]), int(gt[3]), int(gt[4]), list(gt[5]) if gt[5] is not None else None]))
    return out


def run_reference_clusters(ti, extra_args=()):
    """The reference's own cluster.resolve (cluster.py:219-353) on one task, SV type by SV type: the `--dev-dump-clusters`
    BED text (the clusters after the merge scan) and every yielded cluster (after merge_inner / resplit / resplit_bnd)."""
    import contextlib
    import io
    import tempfile
    ref = load_reference()
    cfg = make_config(tuple(extra_args), ti.qc_nm_threshold)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cfg.vcf = os.path.join(tmp, "out.vcf")
        cfg.dev_dump_clusters = True
        task = build_task(ti, cfg)
        lp = task.lead_provider
        for svtype in ref.sv.ALL_TYPES:
            with contextlib.redirect_stdout(io.StringIO()):
                yielded = list(ref.cluster.resolve(svtype, lp, cfg, task.tandem_repeats))
            path = f"{cfg.vcf}.clusters.{svtype}.{lp.contig}.{lp.start}.{lp.end}.bed"
            bed = open(path).read() if os.path.exists(path) else ""
            out[svtype] = dict(bed=bed, yielded=[dict(
                id=c.id, start=int(c.start), end=int(c.end), seed=int(c.seed), repeat=bool(c.repeat),
                leads_long=None if c.leads_long is None else len(c.leads_long),
                leads=[[ld.read_qname, int(ld.ref_start), int(ld.svlen), ld.source] for ld in c.leads]) for c in yielded])
    return out


# ---------------------------------------------------------------------------------------------- combine (multi-sample)
def cand_record(c) -> dict:
    """Everything resolve_block_groups / SVGroup.call read from a per-sample candidate SVCall."""
    gt = c.genotypes.get(0)
    return dict(id=c.id, contig=c.contig, pos=int(c.pos), end=int(c.end), svtype=c.svtype, svlen=int(c.svlen),
                support=int(c.support), qual=int(c.qual), precise=bool(c.precise), fwd=int(c.fwd), rev=int(c.rev),
                filter=c.filter, qc=bool(c.qc), alt=c.alt, sample=int(c.sample_internal_id),
                cov=[int(c.coverage_upstream), int(c.coverage_start), int(c.coverage_center), int(c.coverage_end),
                     int(c.coverage_downstream)],
                gt=None if gt is None else [gt[0], gt[1], int(gt[2]), int(gt[3]), int(gt[4]), list(gt[5])],
                bnd=None if c.bnd_info is None else [c.bnd_info.mate_contig, int(c.bnd_info.mate_ref_start),
                                                    bool(c.bnd_info.is_first), bool(c.bnd_info.is_reverse)])


def group_call_record(c) -> dict:
    return dict(id=c.id, contig=c.contig, pos=int(c.pos), end=int(c.end), svtype=c.svtype, svlen=int(c.svlen),
                support=int(c.support), qual=c.qual, precise=bool(c.precise), fwd=int(c.fwd), rev=int(c.rev),
                filter=c.filter, alt=c.alt,
                cov=[c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream],
                stdev_pos=_f(c.info.get("STDEV_POS")), stdev_len=_f(c.info.get("STDEV_LEN")),
                genotypes={str(k): [v[0], v[1], v[2], v[3], v[4], list(v[5]), v[6]] for k, v in sorted(c.genotypes.items())})


def fake_coverage(pos_mean: float, sample: int) -> int:
    """Deterministic stand-in for the SNF _COVERAGE lookup of non-included samples (parallel.py:536-551)."""
    return (int(pos_mean) + 3 * sample) % 30


def run_reference_combine(sample_tasks, extra_args=(), split=True):
    """Per-sample candidates from the reference's own call_candidates + finalize_candidates, then the reference's
    resolve_block_groups (two chained windows per SV type so that `groups_initial` is exercised) and SVGroup.call."""
    import oracle as oc  # exact DP (C) as the stand-in for edlib.align(...)['editDistance']
    ref = load_reference()
    ref.sv.align = lambda a, b: {"editDistance": oc.edit_distance(a.encode("latin-1"), b.encode("latin-1"))}
    ns = len(sample_tasks)
    cfg = make_config(tuple(extra_args), sample_tasks[0].qc_nm_threshold)
    per_sample = []
    for s, ti in enumerate(sample_tasks):
        cfg_s = make_config((), ti.qc_nm_threshold)
        task = build_task(ti, cfg_s)
        cands = task.call_candidates(False, cfg_s)
        task.finalize_candidates(cands, True, cfg_s)
        keep = []
        for c in cands:
            if c.svtype not in ref.sv.TYPES or c.support < cfg.combine_support_threshold:
                continue
            c.rnames = None            # SNFile.store drops read names (snf.py:97-98)
            c.sample_internal_id = s
            keep.append(c)
        per_sample.append(keep)
    cfg.snf_input_info = [dict(internal_id=s) for s in range(ns)]
    cfg.mode = "combine"
    task = ref.parallel.Task(id=7, sv_id=0, contig=sample_tasks[0].contig, start=0, end=sample_tasks[0].contig_len,
                             config=cfg)
    out = dict(n_samples=ns, problems=[], calls=[])
    for svtype in ref.sv.TYPES:
        bins = {}
        for s in range(ns):
            for c in per_sample[s]:
                if c.svtype == svtype:
                    bins.setdefault(int(c.pos / cfg.combine_min_size) * cfg.combine_min_size, []).append(c)
        if not bins:
            continue
        ordered = [c for b in sorted(bins) for c in bins[b]]
        windows = [ordered[:len(ordered) // 2], ordered[len(ordered) // 2:]] if split and len(ordered) > 3 else [ordered]
        groups = []
        for w in windows:
            before = [[(c.sample_internal_id, c.id) for c in g.candidates] for g in groups]
            state = [dict(pos_mean=g.pos_mean, len_mean=g.len_mean,
                          mate_mean=None if g.bnd_mate_ref_start_mean is None else float(g.bnd_mate_ref_start_mean))
                     for g in groups]
            groups = ref.cluster.resolve_block_groups(svtype, w, groups, cfg)
            out["problems"].append(dict(
                svtype=svtype, cands=[cand_record(c) for c in w], groups_initial=before, groups_initial_state=state,
                groups=[dict(members=[[c.sample_internal_id, c.id] for c in g.candidates], pos_mean=g.pos_mean,
                             len_mean=g.len_mean,
                             mate_mean=None if g.bnd_mate_ref_start_mean is None else float(g.bnd_mate_ref_start_mean))
                        for g in groups]))
        for g in groups:
            for s in set(range(ns)) - g.included_samples:
                g.coverages_nonincluded[s] = fake_coverage(g.pos_mean, s)
        out["calls"].append(dict(svtype=svtype, calls=[group_call_record(c) for c in ref.sv.call_groups(groups, cfg, task)]))
    return out


def run_reference_combine_task(sample_tasks, extra_args=(), with_objects=False, scatter_target=None, align=None, before_execute=None, timing=None,
                               task_id=7):
    """The reference's own CombineTask.execute (parallel.py:444-572) on a synthetic population.

    Per-sample candidates come from the reference's call_candidates + finalize_candidates; they are put into SNF blocks by
    the reference's SNFile.store and SNFile.annotate_block_coverages (in memory: only the gzip/pickle file layer is
    replaced by a stand-in whose read_blocks returns those block dicts).  Returns the inputs (blocks per sample) as
    records and the combined calls in emission order."""
    import tempfile
    import oracle as oc
    ref = load_reference()
    from sniffles import snf as ref_snf
    # edlib.align(a, b)["editDistance"]: the exact DP (goldens), or - `align="myers"`, the baseline of bench.py --config 4 - the bit-parallel
    # algorithm edlib implements (oracle/snf_oracle.c::snf_oracle_edit_distance_myers, pinned to the DP)
    dist_fn = oc.edit_distance_myers if align == "myers" else oc.edit_distance
    ref.sv.align = lambda a, b: {"editDistance": dist_fn(a.encode("latin-1"), b.encode("latin-1"))}
    ns = len(sample_tasks)
    contig, contig_len = sample_tasks[0].contig, sample_tasks[0].contig_len
    cfg = make_config(tuple(extra_args), sample_tasks[0].qc_nm_threshold)
    blocks_per_sample = []
    for s, ti in enumerate(sample_tasks):
        cfg_s = make_config((), ti.qc_nm_threshold)
        task = build_task(ti, cfg_s)
        cands = task.call_candidates(False, cfg_s)
        task.finalize_candidates(cands, True, cfg_s)
        sf = ref_snf.SNFile(cfg_s, False, filename=None)
        for c in cands:
            sf.store(c)                                     # drops rnames, keeps sv.TYPES only (snf.py:91-100)
        sf.annotate_block_coverages(task.lead_provider)
        blocks_per_sample.append(sf.blocks)

    class FakeSNF:
        reqc = False

        def __init__(self, blocks):
            self.blocks = blocks

        def read_header(self):
            pass

        def close(self):
            pass

        def read_blocks(self, ctg, block_index):
            if ctg != contig or block_index not in self.blocks:
                return None
            return [self.blocks[block_index]]

    tmpdir = tempfile.mkdtemp(prefix="snf_fake_")
    fakes = {}
    infos = []
    for s in range(ns):
        fn = os.path.join(tmpdir, f"s{s}.snf")
        open(fn, "wb").close()
        fakes[fn] = FakeSNF(blocks_per_sample[s])
        infos.append(dict(internal_id=s, filename=fn))
    cfg.snf_input_info = infos
    cfg.mode = "combine"
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(ns)]
    cfg.combine_close_handles = False
    cfg.combine_population = None

    class Collector:
        def __init__(self, task, svcalls, count):
            self.calls = []

        def store_calls(self, svcalls):
            self.calls.extend(svcalls)

        def finalize(self):
            pass

    orig = ref.parallel.snf.SNFile
    ref.parallel.snf.SNFile = lambda config, handle, filename=None: (handle.close(), fakes[filename])[1]
    scattered = None
    old_target = ref.parallel.CombineTask.TARGET_WORK_PER_TASK
    try:
        ctask = ref.parallel.CombineTask(id=task_id, sv_id=0, contig=contig, start=0, end=contig_len, config=cfg, result_class=Collector)
        if before_execute is not None:
            before_execute()                 # (a barrier: every worker of a pool holds its samples' blocks)
        import time as _time
        _t0 = _time.perf_counter()
        res = ctask.execute()
        if timing is not None:
            timing["execute_s"] = _time.perf_counter() - _t0
            timing["candidates"] = sum(len(b[t]) for blocks in blocks_per_sample for b in blocks.values() for t in ref.sv.TYPES)
        if scatter_target is not None:
            # the reference's own CombineTask.scatter / clone (parallel.py:411-442) with its class constant lowered so that
            # a test-sized contig is cut (the constant is 10000 blocks x samples), every sub-task executed on its own
            ref.parallel.CombineTask.TARGET_WORK_PER_TASK = scatter_target
            cfg.threads = 4
            ctask2 = ref.parallel.CombineTask(id=7, sv_id=0, contig=contig, start=0, end=contig_len, config=cfg, result_class=Collector)
            scattered = []
            for sub in ctask2.scatter():
                r = sub.execute()
                scattered.append(dict(id=int(sub.id), start=int(sub.start), end=int(sub.end), block_indices=[int(b) for b in sub.block_indices],
                                      calls=[group_call_record(c) for c in r.calls]))
    finally:
        ref.parallel.snf.SNFile = orig
        ref.parallel.CombineTask.TARGET_WORK_PER_TASK = old_target
    inputs = []
    for s in range(ns):
        blk = []
        for bi in sorted(blocks_per_sample[s]):
            b = blocks_per_sample[s][bi]
            recs = {}
            for svtype in ref.sv.TYPES:
                out = []
                for c in b[svtype]:
                    c.sample_internal_id = s
                    out.append(cand_record(c))
                recs[svtype] = out
            blk.append(dict(block=int(bi), cands=recs, coverage={str(k): int(v) for k, v in sorted(b["_COVERAGE"].items())}))
        inputs.append(blk)
    doc = dict(n_samples=ns, contig=contig, contig_len=int(contig_len), samples=inputs,
               calls=[group_call_record(c) for c in res.calls])
    if scattered is not None:
        doc["scatter"] = dict(target_work_per_task=scatter_target, threads=4, tasks=scattered)
    return (doc, res.calls, cfg) if with_objects else doc


# ---------------------------------------------------------------------------------------------- SNF container
def write_reference_snf(ti, path, extra_args=()):
    """The reference's own `.snf` of one task: CallTask.execute's SNF tail (parallel.py:278-295: SNFile.store,
    annotate_block_coverages on the dense coverage vector, write_and_index into a part file) followed by the main
    program's SNFile.write_results (sniffles:269, snf.py:186-223).  Returns the candidate count."""
    load_reference()
    from sniffles import snf as ref_snf
    cfg = make_config(("--snf", path) + tuple(extra_args), ti.qc_nm_threshold)
    cfg.contig_lengths = [(ti.contig, int(ti.contig_len))]
    task = build_task(ti, cfg)
    cands = task.call_candidates(False, cfg)
    task.finalize_candidates(cands, True, cfg)
    part = f"{path}.tmp_{task.id}.snf"
    with open(part, "wb") as handle:
        so = ref_snf.SNFile(cfg, handle)
        for c in cands:
            so.store(c)
        so.annotate_block_coverages(task.lead_provider)
        so.write_and_index()
    res = types.SimpleNamespace(has_snf=True, contig=ti.contig, task_id=task.id, snf_filename=part, snf_index=so.get_index(),
                                snf_total_length=so.get_total_length(), snf_candidate_count=len(cands),
                                coverage_average_total=task.coverage_average_total)
    out = ref_snf.SNFile(cfg, open(path, "wb"))
    out.add_result(res)
    n = out.write_results(cfg, [ti.contig])
    out.close()
    return n


def open_reference_snf(path):
    load_reference()
    from sniffles import snf as ref_snf
    cfg = make_config(())
    cfg.combine_close_handles = False
    f = ref_snf.SNFile(cfg, open(path, "rb"), filename=path)
    f.read_header()
    return f


# ---------------------------------------------------------------------------------------------- VCF writer
def run_reference_call_svs(ti, extra_args=(), overrides=None):
    """CallTask.execute's calling part (parallel.py:255-271) on one task: candidates -> finalize -> QC filter -> sort.
    Returns (calls, config)."""
    cfg = make_config(tuple(extra_args), ti.qc_nm_threshold)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    task = build_task(ti, cfg)
    qc = not (cfg.snf is not None or cfg.no_qc)
    cands = task.call_candidates(qc, cfg)
    calls = task.finalize_candidates(cands, not qc, cfg)
    if not cfg.no_qc:
        calls = [c for c in calls if c.qc]
    if cfg.sort:
        calls = sorted(calls, key=lambda c: c.pos)
    return calls, cfg


def reference_vcf_text(calls, cfg, contigs_lengths, fasta=None) -> str:
    """Header and records as the UNMODIFIED reference writer (vcf.py) emits them for `calls` (deep-copied: the writer
    mutates the calls it writes)."""
    import copy
    import io
    load_reference()
    from sniffles import vcf as ref_vcf
    buf = io.StringIO()
    w = ref_vcf.VCF(cfg, buf)
    w.reference_handle = fasta
    w.write_header(contigs_lengths)
    n = sum(w.write_call(c) for c in copy.deepcopy(calls))
    assert n == w.call_count
    return buf.getvalue()


def reference_vcf_records(calls, cfg, fasta=None) -> str:
    """The record lines alone (no header) of the UNMODIFIED reference writer for `calls`, which it may mutate."""
    import io
    load_reference()
    from sniffles import vcf as ref_vcf
    buf = io.StringIO()
    w = ref_vcf.VCF(cfg, buf)
    w.reference_handle = fasta
    for c in calls:
        w.write_call(c)
    return buf.getvalue()


# ---------------------------------------------------------------------------------------------- whole sample
class DictFasta:
    """pysam.FastaFile stand-in over {contig: sequence} (fetch with pysam's clipping; unknown contig -> KeyError)."""

    def __init__(self, seqs):
        self.seqs = seqs

    def fetch(self, contig, start=None, end=None):
        s = self.seqs[contig]
        return s[(0 if start is None else start):(len(s) if end is None else end)]


def run_reference_call_sample(recs, extra_args=(), snf_path=None, fixed=None, fasta=None):
    """The reference's `call_sample` flow on an in-memory BAM (`sniffles_amd.bam.BamRecords`): the main program's task
    layout (sniffles:286-360; default task_count_multiplier 0 = one task per processed contig), the UNMODIFIED
    `CallTask.execute` (build_leadtab over oracle/pysam_stub, call_candidates, finalize_candidates, SNF part) per task in
    this process, results emitted in task order through the unmodified VCF writer and `SNFile.write_results`.
    `fasta`: {contig: sequence} - the run then has `--reference`: `_mask_N_coverage` (leadprov.py:420-443) and the writer's REF / ALT
    resolution read it through a pysam.FastaFile stand-in.
    Returns dict(vcf=text, read_count=..., snf_candidates=...)."""
    import io
    import math
    import struct
    import pysam_stub
    ref = load_reference()
    from sniffles import snf as ref_snf, vcf as ref_vcf
    args = ["--input", "x.bam", "--vcf", "out.vcf"] + (["--snf", snf_path] if snf_path else []) + list(extra_args)
    cfg = ref.config.SnifflesConfig(*args)
    cfg.mode = "call_sample"
    cfg.input_is_cram, cfg.input_mode = False, "rb"
    cfg.sample_ids_vcf = [(0, "SAMPLE")]
    for k, v in (fixed or {}).items():
        setattr(cfg, k, v)
    flags = [struct.unpack_from("<H", recs.blob, int(o) + 18)[0] for o in recs.rec_off[:-1]]
    total_mapped = sum(1 for f, r in zip(flags, recs.ref_id.tolist()) if r >= 0 and not f & 0x4)
    cfg.task_read_id_offset_mult = 10 ** 9 if total_mapped == 0 else 10 ** math.ceil(math.log(total_mapped) + 1)
    contig_lengths = [(c, int(n)) for c, n in zip(recs.ref_names, recs.ref_lens) if ref.util.should_process_contig(c, int(n), cfg)]
    cfg.contig_lengths = contig_lengths
    buf = io.StringIO()
    vcf_out = ref_vcf.VCF(cfg, buf)
    vcf_out.write_header(contig_lengths)
    snf_out = ref_snf.SNFile(cfg, open(snf_path, "wb")) if snf_path else None
    orig = ref.parallel.pysam.AlignmentFile
    ref.parallel.pysam.AlignmentFile = lambda *a, **k: pysam_stub.AlignmentFile(recs)
    orig_fasta = ref.leadprov.pysam.FastaFile
    if fasta is not None:
        cfg.reference = "reference.fa"
        ref.leadprov.pysam.FastaFile = lambda *a, **k: DictFasta(fasta)
        vcf_out.reference_handle = DictFasta(fasta)      # (VCF.open_reference, vcf.py:108-120, without the file system)
    read_count = 0
    try:
        for task_id, (contig, length) in enumerate(contig_lengths):
            task = ref.parallel.CallTask(id=task_id, contig=contig, start=0, end=length - 1, assigned_process_id=None,
                                         tandem_repeats=(getattr(recs, "tandem_repeats", None) or {}).get(contig),
                                         genotype_svs=None, sv_id=0, config=cfg,
                                         regions=(getattr(cfg, "regions_by_contig", None) or {}).get(contig))   # sniffles:351
            result = task.execute()
            read_count += result.processed_read_count
            result.emit(vcf_out=vcf_out, snf_out=snf_out)
    finally:
        ref.parallel.pysam.AlignmentFile = orig
        ref.leadprov.pysam.FastaFile = orig_fasta
    n_snf = 0
    if snf_out is not None:
        n_snf = snf_out.write_results(cfg, [c for c, _ in contig_lengths])
        snf_out.close()
    return dict(vcf=buf.getvalue(), read_count=int(read_count), snf_candidates=int(n_snf), contig_lengths=contig_lengths)


# ---------------------------------------------------------------------------------------------- force calling
def run_reference_genotype(ti, specs, extra_args=()):
    """The UNMODIFIED reference GenotypeTask.execute (parallel.py:299-372) on one task with the target SVs `specs`
    (tests/genotype_util.py); build_leadtab is replaced by the prepared LeadProvider.  Returns the per-target records
    (matched candidate id, distance, coverage samples, genotype) or dict(error=...)."""
    import genotype_util as gutil
    ref = load_reference()
    cfg = make_config(tuple(extra_args), ti.qc_nm_threshold)
    base = build_task(ti, cfg)

    def new_call(cls):
        return cls(contig=None, pos=0, id="", ref="N", alt="", qual=0, filter="PASS", info=dict(), svtype="", svlen=0, end=0,
                   genotypes=dict(), precise=False, support=0, rnames=None, qc=True, nm=-1, postprocess=None)
    targets = gutil.make_targets(specs, ref.sv.SVCall, ref.sv.SVCallBNDInfo, new_call)
    task = ref.parallel.GenotypeTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                                     genotype_svs=targets)
    task.lead_provider, task.tandem_repeats = base.lead_provider, base.tandem_repeats
    task.build_leadtab = lambda: (None, 0)
    try:
        res = task.execute()
    except Exception as e:
        return dict(error=type(e).__name__)
    if res is None:
        return dict(returned_none=True)
    return dict(targets=gutil.result_records(res.svcalls))


def run_reference_population(recs_list, workdir, extra_args=(), fixed=None):
    """BAMs -> per-sample .snf (run_reference_call_sample) -> the reference's `combine` flow (sniffles:371-490 in this
    process): headers, one unmodified CombineTask.execute per contig reading the real files (edlib replaced by the
    exact DP, see the module header), calls sorted per task, the unmodified VCF writer.  Returns dict(vcf=text, snf=[paths])."""
    import io
    import oracle as oc
    ref = load_reference()
    from sniffles import snf as ref_snf, vcf as ref_vcf
    ref.sv.align = lambda a, b: {"editDistance": oc.edit_distance(a.encode("latin-1"), b.encode("latin-1"))}
    paths = []
    for s, recs in enumerate(recs_list):
        path = os.path.join(workdir, f"sample{s}.snf")
        run_reference_call_sample(recs, (), path, fixed)
        paths.append(path)
    cfg = ref.config.SnifflesConfig("--input", *paths, "--vcf", "out.vcf", *extra_args)
    cfg.mode = "combine"
    for k, v in (fixed or {}).items():
        setattr(cfg, k, v)
    cfg.snf_input_info, cfg.sample_ids_vcf = [], []
    contig_lengths = None
    for internal_id, path in enumerate(paths):
        f = ref_snf.SNFile(cfg, open(path, "rb"), filename=path)
        f.read_header()
        contig_lengths = f.header["config"]["contig_lengths"]
        sid = f.header["config"]["sample_id"] or os.path.splitext(os.path.basename(path))[0]
        cfg.snf_input_info.append({"internal_id": internal_id, "sample_id": sid, "filename": path})
        f.close()
    cfg.sample_ids_vcf = [(i["internal_id"], i["sample_id"]) for i in cfg.snf_input_info]
    cfg.combine_close_handles = False
    buf = io.StringIO()
    w = ref_vcf.VCF(cfg, buf)
    w.write_header(contig_lengths)

    class Collector:
        def __init__(self, task, svcalls, count):
            self.calls = []

        def store_calls(self, svcalls):
            self.calls.extend(svcalls)

        def finalize(self):
            pass
    for task_id, (contig, length) in enumerate(contig_lengths):
        task = ref.parallel.CombineTask(id=task_id, sv_id=0, contig=contig, start=0, end=length - 1, assigned_process_id=None,
                                        config=cfg, result_class=Collector, regions=None)
        res = task.execute()
        for c in sorted(res.calls, key=lambda c: c.pos):
            w.write_call(c)
    return dict(vcf=buf.getvalue(), snf=paths)


def run_reference_genotype_vcf(recs, vcf_text, extra_args=(), fixed=None):
    """The reference's `--genotype-vcf` flow in this process: its VCF reader on `vcf_text`, one unmodified
    GenotypeTask.execute per processed contig (build_leadtab over oracle/pysam_stub), GenotypeResult.emit through the
    unmodified rewriter.  Returns the output VCF text."""
    import io
    import math
    import struct
    import pysam_stub
    ref = load_reference()
    from sniffles import vcf as ref_vcf
    cfg = ref.config.SnifflesConfig("--input", "x.bam", "--vcf", "out.vcf", "--genotype-vcf", "in.vcf", *extra_args)
    cfg.mode = "genotype_vcf"
    cfg.input_is_cram, cfg.input_mode = False, "rb"
    for k, v in (fixed or {}).items():
        setattr(cfg, k, v)
    vcf_in = ref_vcf.VCF(cfg, io.StringIO(vcf_text))
    order, by_contig = [], {}
    for svcall in vcf_in.read_svs_iter():
        order.append(svcall.raw_vcf_line_index)
        by_contig.setdefault(svcall.contig, []).append(svcall)
    flags = [struct.unpack_from("<H", recs.blob, int(o) + 18)[0] for o in recs.rec_off[:-1]]
    total_mapped = sum(1 for f, r in zip(flags, recs.ref_id.tolist()) if r >= 0 and not f & 0x4)
    cfg.task_read_id_offset_mult = 10 ** 9 if total_mapped == 0 else 10 ** math.ceil(math.log(total_mapped) + 1)
    contig_lengths = [(c, int(n)) for c, n in zip(recs.ref_names, recs.ref_lens) if ref.util.should_process_contig(c, int(n), cfg)]
    cfg.contig_lengths = contig_lengths
    buf = io.StringIO()
    vcf_out = ref_vcf.VCF(cfg, buf)
    vcf_out.rewrite_header_genotype(vcf_in.header_str)
    orig = ref.parallel.pysam.AlignmentFile
    ref.parallel.pysam.AlignmentFile = lambda *a, **k: pysam_stub.AlignmentFile(recs)
    try:
        for task_id, (contig, length) in enumerate(contig_lengths):
            targets = [t for t in by_contig.get(contig, []) if 0 <= t.pos < length - 1]
            task = ref.parallel.GenotypeTask(id=task_id, contig=contig, start=0, end=length - 1, assigned_process_id=None,
                                             tandem_repeats=(getattr(recs, "tandem_repeats", None) or {}).get(contig),
                                             genotype_svs=targets, sv_id=0, config=cfg,
                                             regions=(getattr(cfg, "regions_by_contig", None) or {}).get(contig))   # sniffles:351
            result = task.execute()
            if result is not None:
                result.emit(vcf_out=vcf_out, genotype_lineindex_order=order)
    finally:
        ref.parallel.pysam.AlignmentFile = orig
    return buf.getvalue()


# ---------------------------------------------------------------------------------------------- signature extraction
def lead_record(ld) -> list:
    """Canonical JSON-able row of one reference Lead as `record_lead` receives it (before the per-bin seq cap)."""
    b = ld.bnd_info
    return [int(ld.read_id), ld.read_qname, ld.contig, int(ld.ref_start), int(ld.ref_end), int(ld.qry_start),
            int(ld.qry_end), ld.strand, int(ld.mapq), None if ld.nm is None else float(ld.nm).hex(), ld.source,
            ld.svtype, None if ld.svlen is None else int(ld.svlen), ld.seq, ld.hap, ld.phase_set, bool(ld.is_sa),
            int(ld.read_len),
            None if b is None else [b.mate_contig, int(b.mate_ref_start), bool(b.is_first), bool(b.is_reverse)]]


def run_reference_extract(recs, contig, start, end, extra_args=(), read_id_offset=0, overrides=None) -> dict:
    """`LeadProvider.build_leadtab([Region(contig, start, end)], bam)` of the UNMODIFIED reference
    (`leadprov.py:445-472` -> `iter_region`, `read_iterindels`, `read_itersplits`, `Lead.for_bnd`,
    `sv.classify_splits`) over `oracle/pysam_stub` objects decoded from the raw records `recs`
    (`sniffles_amd.bam.BamRecords`).  Returns the leads in `record_lead` order, the coverage vector as a sparse
    difference array, the REF haplotype bin counters, the read counters and the NM side channel; or
    dict(error=ExceptionName) when the reference raises."""
    import pysam_stub
    ref = load_reference()
    cfg = ref.config.SnifflesConfig("--input", "x.bam", "--vcf", "out.vcf", *extra_args)
    cfg.mode = "call_sample"
    for k, val in (overrides or {}).items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, val)
    lp = ref.leadprov.LeadProvider(cfg, read_id_offset, contig)
    leads = []
    orig = lp.record_lead

    def rec(ld, pos_leadtab):
        leads.append(lead_record(ld))
        orig(ld, pos_leadtab)
    lp.record_lead = rec
    bam = pysam_stub.AlignmentFile(recs)
    try:
        externals = lp.build_leadtab([ref.leadprov.Region(contig, start, end)], bam)
    except Exception as e:
        return dict(error=type(e).__name__)
    cov = lp.coverage.astype(np.int64)
    d = np.diff(cov, prepend=0)
    nz = np.nonzero(d)[0]
    hapref = sorted([int(k)] + [int(x) for x in v] for k, v in lp.leadhapcount["REF"].items())
    return dict(leads=leads, n_externals=len(externals), cov_pos=nz.tolist(), cov_delta=d[nz].tolist(),
                contig_len=int(cov.shape[0]), hapref=hapref, read_count=int(lp.read_count), read_id=int(lp.read_id),
                qc_nm_threshold=float(cfg.qc_nm_threshold).hex(),
                lead_counts={k: int(v) for k, v in lp.leadcounts.items()})
