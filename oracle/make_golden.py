"""TEST INFRASTRUCTURE ONLY - regenerates tests/golden/*.json.gz from the UNMODIFIED reference.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
Each fixture holds: the case name, a SHA-256 over the input arrays (so the test notices if the
seeded generators drift), the reference's candidate-stage and final-stage records, and
coverage_average_total.  Inputs are rebuilt at test time from tests/cases.py.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def input_sha(ti) -> str:
    h = hashlib.sha256()
    for name in sorted(ti.leads):
        h.update(name.encode())
        h.update(ti.leads[name].tobytes())
    for a in (ti.seq_pool, ti.read_start, ti.read_end, ti.read_hp):
        h.update(a.tobytes())
    if ti.tr_start is not None:
        h.update(ti.tr_start.tobytes())
        h.update(ti.tr_end.tobytes())
    h.update(repr((ti.task_id, ti.contig, ti.contig_len, ti.sv_id_start, ti.qc_nm_threshold)).encode())
    return h.hexdigest()


def main(names=None):
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (build, kw, args) in cases.ALL.items():
        if names and name not in names:
            continue
        ti = build()
        ref = rh.run_reference(ti, args)
        doc = dict(case=name, config=kw, reference_args=list(args), input_sha=input_sha(ti), expected=ref)
        path = os.path.join(out_dir, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        n = "error:" + ref["error"] if "error" in ref else f"{len(ref['candidates'])} cand / {len(ref['final'])} final"
        print(f"{name:32s} {ti.n_leads:7d} leads  {n}")




CLUSTER_CASES = ["chr20_30x_ont", "chr21_30x_mosaic", "chr19_30x_no_tr", "merge_inner", "resplit_wrap", "merge_index_rule", "tr_sweep_repeat",
                 "bnd_stale_end", "long_ins", "fuzz_0_0", "fuzz_3_3", "fuzz_5_4", "fuzz_10_5"]


def main_clusters(names=None):
    """Seam B3: the reference's cluster.resolve per SV type (dump-point BED text + yielded clusters) for a few of the cases."""
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    doc = {}
    for name in CLUSTER_CASES:
        build, kw, args = cases.ALL[name]
        ti = build()
        doc[name] = dict(input_sha=input_sha(ti), config=kw, reference_args=list(args), clusters=rh.run_reference_clusters(ti, args))
        print(f"{name:32s} " + " ".join(f"{t}:{len(v['yielded'])}" for t, v in doc[name]["clusters"].items()))
    with gzip.GzipFile(os.path.join(out_dir, "clusters_resolve.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


REGENOTYPE_CASES = ["chr20_30x_ont", "chr22_60x_hifi", "phase_rescue", "gt_failed_edges", "long_del_dup_cov", "fuzz_3_0", "fuzz_6_0"]


def main_regenotype():
    """--reqc: the reference's genotype_sv applied once more to finalized candidates."""
    import cases
    import ref_harness as rh
    doc = {}
    for name in REGENOTYPE_CASES:
        build, kw, args = cases.ALL[name]
        ti = build()
        doc[name] = dict(input_sha=input_sha(ti), config=kw, calls=rh.run_reference_regenotype(ti, args))
        print(f"{name:32s} {len(doc[name]['calls'])} candidates re-genotyped")
    with gzip.GzipFile(os.path.join(ROOT, "tests", "golden", "regenotype.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


def main_combine(names=None):
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, args) in cases.COMBINE.items():
        if names and name not in names:
            continue
        tis = build()
        ref = rh.run_reference_combine(tis, args)
        doc = dict(case=name, reference_args=list(args), input_sha=[input_sha(t) for t in tis], expected=ref)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        print(f"{name:32s} {len(tis)} samples  {sum(len(p['cands']) for p in ref['problems'])} cands  "
              f"{sum(len(p['groups']) for p in ref['problems'])} groups  {sum(len(c['calls']) for c in ref['calls'])} calls")




def main_consensus():
    """Golden vectors for seam B4: the reference's own consensus.novel_from_reads on seeded random problems."""
    import numpy as np
    import ref_harness as rh
    ref = rh.load_reference()
    rng = np.random.default_rng(20260924)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    odd = np.frombuffer(b"ACGTNacgtR", dtype=np.uint8)

    class L:  # only .seq is read
        def __init__(self, seq):
            self.seq = seq

    def mutate(a, rate, indel=0.06):
        # mostly substitutions: the reference only uses anchors within 6 positions of the diagonal
        out = []
        for ch in a:
            u = rng.random()
            if u < rate * indel:
                continue
            if u < 2 * rate * indel:
                out.append(int(alpha[rng.integers(4)]))
            out.append(int(alpha[rng.integers(4)]) if u < rate else int(ch))
        return bytes(out)

    cases_ = []
    for k in range(160):
        Lb = int(rng.choice([7, 20, 46, 90, 150, 320, 499, 520, 800, 1500, 3000, 6100]))
        if k % 17 == 0:
            Lb = int(rng.integers(1, 40))
        base = alpha[rng.integers(0, 4, Lb)]
        if k % 9 == 0:   # low complexity: repeated k-mers (taboo anchors)
            base = np.tile(alpha[rng.integers(0, 4, 5)], Lb // 5 + 1)[:Lb]
        true = bytes(base.tolist())
        best = mutate(true, float(rng.choice([0.0, 0.03, 0.08]))) or true   # the best read has errors of its own
        Lb = len(best)
        n_others = int(rng.choice([0, 1, 2, 4, 5, 8, 12, 20, 40]))
        rate = float(rng.choice([0.0, 0.02, 0.05, 0.15]))
        others = []
        for _ in range(n_others):
            o = mutate(true, rate)
            if rng.random() < 0.15:   # unrelated read
                o = bytes(alpha[rng.integers(0, 4, max(1, int(Lb * rng.uniform(0.5, 1.5))))].tolist())
            if rng.random() < 0.1:    # odd characters
                o = bytes(int(odd[rng.integers(odd.shape[0])]) if rng.random() < 0.01 else c for c in o)
            if rng.random() < 0.1:
                o = o[: max(1, len(o) // 2)]
            others.append(o)
        klen = 6
        skip = 3 + int(Lb * (1.0 / 500.0))
        if k % 13 == 0:
            skip = int(rng.integers(1, 9))
            if Lb / skip > 450:      # stay inside the limits of the workgroup kernels (500 sampled positions)
                skip = 3 + int(Lb * (1.0 / 500.0))
        exp = ref.consensus.novel_from_reads(L(best.decode("latin-1")), [L(o.decode("latin-1")) for o in others], klen=klen,
                                             skip=skip, skip_repetitive=skip)
        cases_.append(dict(best=best.decode("latin-1"), others=[o.decode("latin-1") for o in others], klen=klen, skip=skip,
                           expected=exp))
    doc = dict(case="consensus_novel_from_reads", n=len(cases_), problems=cases_)
    with gzip.GzipFile(os.path.join(ROOT, "tests", "golden", "consensus_novel_from_reads.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
    changed = sum(1 for c in cases_ if c["expected"] != c["best"])
    print(f"consensus_novel_from_reads: {len(cases_)} problems, {changed} with a consensus that differs from the best read")


def main_consensus_long():
    """Golden vectors for seam B4 with LONG copied segments (anchors far apart): stretches of an other read in which every
    sampled k-mer carries a substitution (the stretch stays > 50 % identical: copied as one segment), stretches that leave
    the +-6 shift window and come back to the same shift (not copied), the same with odd characters - from the reference's
    own consensus.novel_from_reads."""
    import numpy as np
    import ref_harness as rh
    ref = rh.load_reference()
    rng = np.random.default_rng(20260925)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)

    class L:
        def __init__(self, seq):
            self.seq = seq

    def other_base(c):
        return int(alpha[(int(np.searchsorted(alpha, c)) + int(rng.integers(1, 4))) % 4])

    cases_ = []
    for k in range(60):
        Lb = int([300, 380, 1500, 3000, 6100, 8000][k % 6])
        skip = 3 + int(Lb * (1.0 / 500.0))
        if k >= 36:     # sampling steps longer than the 24-byte register window of the kernels (consensus_kmer_skip_base is free)
            skip = int([24, 31, 40, 23][k % 4])
        truth = alpha[rng.integers(0, 4, Lb)].copy()
        best = truth.copy()
        err = rng.random(Lb) < float(rng.choice([0.0, 0.02, 0.05]))
        best[err] = [other_base(c) for c in best[err]]
        others = []
        for r in range(int(rng.choice([3, 5, 9, 14]))):
            o = truth.copy()
            sub = rng.random(Lb) < 0.01
            o[sub] = [other_base(c) for c in o[sub]]
            o = bytearray(o.tobytes())
            kind = int(rng.integers(0, 5))
            a = int(rng.integers(0, max(1, Lb // 2)))
            b = min(Lb - 12, a + int(rng.choice([60, 100, 200, 520, 1100, 2600])))
            if kind <= 2 and b > a + 20:
                # every k-mer sampled inside [a, b) broken by one substitution (kind 2: two per sampling step where the step allows)
                for j in range((a // skip + 1) * skip, b, skip):
                    o[j + 2] = other_base(o[j + 2])
                    if kind == 2 and skip >= 8:
                        o[j + 5] = other_base(o[j + 5])
            elif kind == 3 and b > a + 40:
                ins = bytes(alpha[rng.integers(0, 4, 8)].tolist())      # shift +8 inside [a, b), back to 0 behind it
                o = o[:a] + ins + o[a:b] + o[b + 8:]
            if rng.random() < 0.2:
                for j in rng.integers(0, len(o), 6):
                    o[int(j)] = int(rng.choice(list(b"Nnacgt")))
            others.append(bytes(o))
        bs = best.tobytes().decode("latin-1")
        exp = ref.consensus.novel_from_reads(L(bs), [L(o.decode("latin-1")) for o in others], klen=6, skip=skip, skip_repetitive=skip)
        cases_.append(dict(best=bs, others=[o.decode("latin-1") for o in others], klen=6, skip=skip, expected=exp))
    doc = dict(case="consensus_long_segments", n=len(cases_), problems=cases_)
    with gzip.GzipFile(os.path.join(ROOT, "tests", "golden", "consensus_long_segments.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
    changed = sum(1 for c in cases_ if c["expected"] != c["best"])
    print(f"consensus_long_segments: {len(cases_)} problems, {changed} with a consensus that differs from the best read")


def main_combine_task(names=None):
    """Goldens for the CombineTask.execute driver: inputs (SNF blocks per sample) and the combined calls it emits."""
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, args) in cases.COMBINE_TASK.items():
        if names and name not in names:
            continue
        tis = build()
        # one case also pins CombineTask.scatter (parallel.py:422-442): sub-tasks of consecutive blocks, each executed alone
        ref = rh.run_reference_combine_task(tis, args, scatter_target=40 if name == "combine_task_6samples" else None)
        doc = dict(case=name, reference_args=list(args), input_sha=[input_sha(t) for t in tis], expected=ref)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        ncand = sum(len(v) for smp in ref["samples"] for blk in smp for v in blk["cands"].values())
        print(f"{name:32s} {ref['n_samples']} samples  {ncand} candidates in SNF blocks -> {len(ref['calls'])} combined calls")


def main_bam_fixtures():
    """tests/golden/bam_hg008.bam.gz / bam_hg002.bam.gz: the reference's two test BAMs (src/tests/data) as inflated BAM
    streams with the base qualities blanked (0xFF = absent; nothing on the path reads them), gzip-compressed."""
    import struct
    from sniffles_amd import bam
    for name in ("hg008", "hg002"):
        with open(f"/root/reference/src/tests/data/{name}.bam", "rb") as f:
            raw = bytearray(bam.bgzf_inflate(f.read()))
        recs = bam.parse_bam(bytes(raw))
        start = len(raw) - recs.blob.shape[0]
        for i in range(recs.n):
            o = start + int(recs.rec_off[i])
            l_name, n_cig, l_seq = raw[o + 12], struct.unpack_from("<H", raw, o + 16)[0], struct.unpack_from("<i", raw, o + 20)[0]
            q = o + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2
            raw[q:q + l_seq] = b"\xff" * l_seq
        path = os.path.join(ROOT, "tests", "golden", f"bam_{name}.bam.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(bytes(raw))
        print(f"{path}: {recs.n} records, {os.path.getsize(path)} bytes")


def records_sha(recs) -> str:
    h = hashlib.sha256()
    h.update(recs.blob.tobytes())
    h.update(recs.rec_off.tobytes())
    h.update(repr((recs.ref_names, recs.ref_lens)).encode())
    return h.hexdigest()


def main_extract(names=None):
    """Goldens for signature extraction: what the unmodified reference's build_leadtab produces for raw BAM records."""
    import cases
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, case in cases.EXTRACT.items():
        if names and name not in names:
            continue
        recs = cases.extract_records(case)
        ref = rh.run_reference_extract(recs, case["contig"], case["region"][0], case["region"][1], case["args"],
                                       case["read_id_offset"], case["overrides"])
        doc = dict(case=name, reference_args=list(case["args"]), overrides=case["overrides"], input_sha=records_sha(recs),
                   expected=ref)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        if "error" in ref:
            print(f"{name:28s} {recs.n:5d} records  error:{ref['error']}")
        else:
            print(f"{name:28s} {recs.n:5d} records  {ref['read_count']:5d} reads accepted  {len(ref['leads']):6d} leads "
                  f"{ref['lead_counts']}")


def main_snf():
    """SNF container fixtures: real `.snf` files written by the unmodified reference (SNFile.store /
    annotate_block_coverages / write_and_index / write_results) for the samples of cases.SNF_FILES, committed under
    tests/golden/ together with their canonical records (tests/snf_util.py)."""
    import cases
    import ref_harness as rh
    import snf_util as su
    ref = rh.load_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, per_sample_args) in cases.SNF_FILES.items():
        tis = build()
        files = []
        for s, ti in enumerate(tis):
            path = os.path.join(out_dir, f"{name}_s{s}.snf")
            n = rh.write_reference_snf(ti, path, per_sample_args[s])
            f = rh.open_reference_snf(path)
            rec = su.file_record(f, ti.contig, ref.sv.TYPES)
            f.close()
            files.append(dict(file=os.path.basename(path), sha256=su.sha(path), args=list(per_sample_args[s]), record=rec))
            print(f"{path}: {n} candidates, {len(rec['blocks'])} blocks, {os.path.getsize(path)} bytes")
        doc = dict(case=name, input_sha=[input_sha(t) for t in tis], files=files)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


def main_vcf():
    """VCF text written by the unmodified reference writer: single-sample calls of cases.ALL (vcf_util.CASES x
    vcf_util.VARIANTS) and the combined calls of a CombineTask golden (multi-sample columns, AC / SUPP_VEC)."""
    import cases
    import ref_harness as rh
    import vcf_util as vu
    out_dir = os.path.join(ROOT, "tests", "golden")
    doc = dict(single={}, combine={})
    for name in vu.CASES:
        build, kw, args = cases.ALL[name]
        ti = build()
        per = {}
        for vname, (vargs, overrides, with_fasta) in vu.VARIANTS.items():
            calls, cfg = rh.run_reference_call_svs(ti, tuple(args) + tuple(vargs), {k: v for k, v in overrides.items() if k != "symbolic"})
            for k, v in vu.FIXED.items():
                setattr(cfg, k, v)
            cfg.sample_ids_vcf = [(0, "SAMPLE")]
            fasta = vu.FakeFasta({ti.contig: ti.contig_len}) if with_fasta else None
            per[vname] = rh.reference_vcf_text(calls, cfg, [(ti.contig, ti.contig_len)], fasta)
        doc["single"][name] = dict(input_sha=input_sha(ti), text=per)
        print(f"{name:24s}", {k: len(vu.split_text(v)[1]) for k, v in per.items()})
    for name in ("combine_task_3samples_lowcov", "combine_task_8samples_dense"):
        build, args = cases.COMBINE_TASK[name]
        tis = build()
        _, calls, cfg = rh.run_reference_combine_task(tis, args, with_objects=True)
        for k, v in vu.FIXED.items():
            setattr(cfg, k, v)
        calls = sorted(calls, key=lambda c: c.pos)          # CombineResult.store_calls / finalize (result.py:137-147)
        per = {}
        for vname, with_fasta in (("plain", False), ("fasta", True)):
            fasta = vu.FakeFasta({tis[0].contig: tis[0].contig_len}) if with_fasta else None
            per[vname] = rh.reference_vcf_text(calls, cfg, [(tis[0].contig, tis[0].contig_len)], fasta)
        doc["combine"][name] = dict(text=per)
        print(f"{name:32s}", {k: len(vu.split_text(v)[1]) for k, v in per.items()})
    with gzip.GzipFile(os.path.join(out_dir, "vcf_text.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


def main_sample(names=None):
    """End to end: a synthetic coordinate-sorted BAM (cases.SAMPLES) through the unmodified reference's call_sample flow
    (ref_harness.run_reference_call_sample) -> the VCF text and the content of the SNF file it writes."""
    import tempfile
    import cases
    import ref_harness as rh
    import snf_util as su
    import vcf_util as vu
    ref = rh.load_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, args) in {**cases.SAMPLES, **cases.SAMPLES_EMU}.items():
        if names and name not in names:
            continue
        recs = build()
        plain = rh.run_reference_call_sample(recs, args, None, vu.FIXED)
        path = os.path.join(tempfile.mkdtemp(prefix="snf_e2e_"), "sample.snf")
        with_snf = rh.run_reference_call_sample(recs, args, path, vu.FIXED)
        f = rh.open_reference_snf(path)
        snf_rec = {c: su.file_record(f, c, ref.sv.TYPES) for c, _ in plain["contig_lengths"]}
        f.close()
        doc = dict(case=name, reference_args=list(args), input_sha=records_sha(recs), vcf=plain["vcf"], vcf_with_snf=with_snf["vcf"],
                   read_count=plain["read_count"], snf_candidates=with_snf["snf_candidates"], snf=snf_rec)
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as fh:
            fh.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        print(f"{name:24s} {recs.n} records, {plain['read_count']} reads accepted -> {len(vu.split_text(plain['vcf'])[1])} VCF records, "
              f"{with_snf['snf_candidates']} SNF candidates")


def main_genotype():
    """Force calling: the reference's GenotypeTask.execute on cases of cases.ALL with targets derived from the case's own
    candidate records (tests/genotype_util.py)."""
    import cases
    import genotype_util as gutil
    import ref_harness as rh
    out_dir = os.path.join(ROOT, "tests", "golden")
    doc = {}
    for i, name in enumerate(gutil.CASES):
        build, kw, args = cases.ALL[name]
        ti = build()
        with gzip.open(os.path.join(out_dir, name + ".json.gz"), "rb") as f:
            cands = json.loads(f.read().decode())["expected"]["candidates"]
        specs = gutil.target_specs(cands, ti.contig_len, 700 + i)
        doc[name] = dict(input_sha=input_sha(ti), specs=specs, expected=rh.run_reference_genotype(ti, specs, args))
        exp = doc[name]["expected"]
        print(f"{name:24s} {len(specs)} targets", "error " + exp["error"] if "error" in exp else
              f"{sum(1 for t in exp['targets'] if t['match'])} matched")
    # the reference's own failure: a BND target before any other one
    build, kw, args = cases.ALL["bnd_stale_end"]
    ti = build()
    specs = [dict(id="T0", svtype="BND", pos=1000, svlen=0, bnd=["chr2", 5, True, False], cov=[0] * 5),
             dict(id="T1", svtype="DEL", pos=2000, svlen=-100, bnd=None, cov=[0] * 5)]
    doc["bnd_first"] = dict(input_sha=input_sha(ti), specs=specs, expected=rh.run_reference_genotype(ti, specs, args), case="bnd_stale_end")
    print("bnd_first", doc["bnd_first"]["expected"])
    with gzip.GzipFile(os.path.join(out_dir, "genotype_targets.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


def main_population(names=None):
    """BAMs -> per-sample SNF -> merged multi-sample VCF, everything by the unmodified reference (cases.POPULATIONS)."""
    import tempfile
    import cases
    import ref_harness as rh
    import vcf_util as vu
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (build, args) in cases.POPULATIONS.items():
        if names and name not in names:
            continue
        recs = build()
        res = rh.run_reference_population(recs, tempfile.mkdtemp(prefix="pop_"), args, vu.FIXED)
        doc = dict(case=name, reference_args=list(args), input_sha=[records_sha(r) for r in recs], vcf=res["vcf"])
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as fh:
            fh.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
        print(f"{name:28s} {len(recs)} samples -> {len(vu.split_text(res['vcf'])[1])} merged VCF records")


def main_genotype_vcf():
    """Force calling end to end: a target VCF (the reference's own calls for a sample, perturbed, plus hand-made records) and
    the sample's BAM through the reference's --genotype-vcf flow (ref_harness.run_reference_genotype_vcf)."""
    import numpy as np
    import cases
    import ref_harness as rh
    import vcf_util as vu
    out_dir = os.path.join(ROOT, "tests", "golden")
    doc = {}
    for name, seed in (("sample_splits_14x", 3), ("sample_two_contigs_12x", 4)):
        with gzip.open(os.path.join(out_dir, name + ".json.gz"), "rb") as f:
            sample = json.loads(f.read().decode())
        recs = cases.SAMPLES[name][0]()
        hdr, body = vu.split_text(sample["vcf"])
        rng = np.random.default_rng(seed)
        lines = []
        for ln in body:
            f = ln.split("\t")
            if rng.random() < 0.15:
                continue
            f[1] = str(max(1, int(f[1]) + int(rng.choice([0, 0, 3, -40, 700]))))
            lines.append("\t".join(f[:8]))           # a site list: no FORMAT / sample columns
        lines += ["chr20\t500000\tfar\tN\t<DEL>\t.\tPASS\tSVTYPE=DEL;SVLEN=-300;END=500300",
                  "chr20\t600001\tseqins\tA\tAGGGTTTCCCAAAGGGTTTCCCAAAGGGTTTCCCAAAGGGTTTCCCAAAGGGTTTCCCAAAGG\t30\tPASS\tPRECISE",
                  "chrM_short\t100\tshort\tN\t<DEL>\t.\tPASS\tSVTYPE=DEL;SVLEN=-100",
                  "chr21\t700000\ttra\tN\tN[chr20:12345[\t.\tPASS\tSVTYPE=TRA"]
        hdr = [h for h in hdr if not h.startswith("##FORMAT=<ID=GQ") and not h.startswith("##FORMAT=<ID=DV")]
        text = "\n".join(hdr + lines) + "\n"
        out = rh.run_reference_genotype_vcf(recs, text, (), vu.FIXED)
        doc[name] = dict(input_sha=records_sha(recs), vcf_in=text, vcf_out=out)
        print(f"{name:24s} {len(lines)} targets -> {len(vu.split_text(out)[1])} records written")
    with gzip.GzipFile(os.path.join(out_dir, "genotype_vcf.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())


if __name__ == "__main__":
    # python oracle/make_golden.py                 -> every fixture family
    # python oracle/make_golden.py vcf sample      -> only these families
    # python oracle/make_golden.py main fuzz_4_2   -> single cases of the `main` / `combine` families
    FAMILIES = dict(main=main, clusters=main_clusters, regenotype=main_regenotype, combine=main_combine, consensus=main_consensus, consensus_long=main_consensus_long, combine_task=main_combine_task,
                    bam=main_bam_fixtures, extract=main_extract, snf=main_snf, vcf=main_vcf, sample=main_sample, genotype=main_genotype, population=main_population, genotype_vcf=main_genotype_vcf)
    argv = sys.argv[1:]
    fams = [a for a in argv if a in FAMILIES] or list(FAMILIES)
    names = set(a for a in argv if a not in FAMILIES)
    for fam in fams:
        fn = FAMILIES[fam]
        if fam == "clusters":
            fn()
        elif fam in ("main", "combine", "combine_task", "extract", "sample", "population"):
            fn(names or None)
        else:
            fn()
