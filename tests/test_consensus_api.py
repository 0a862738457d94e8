"""Seam B4: sniffles_amd.consensus.novel_from_reads against vectors produced by the UNMODIFIED reference
function (tests/golden/consensus_novel_from_reads.json.gz, oracle/make_golden.py::main_consensus)."""
import pytest

import golden_util as gu


def load():
    return gu.load("consensus_novel_from_reads")["problems"]


def test_golden_vectors_are_meaningful():
    probs = load()
    assert len(probs) >= 100
    assert sum(p["expected"] != p["best"] for p in probs) >= 20          # the consensus really edits the best read
    assert all(len(p["expected"]) == len(p["best"]) for p in probs)      # substitutions only (consensus.py:365-380)


@pytest.mark.gpu
def test_novel_from_reads_batch_matches_reference():
    from sniffles_amd import consensus
    probs = load()
    got = consensus.novel_from_reads_batch([(p["best"], p["others"], p["skip"]) for p in probs], klen=6)
    bad = [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]]
    assert bad == []


@pytest.mark.gpu
def test_long_copied_segments_match_reference():
    """Segments of hundreds to thousands of bases between two anchors (processed by the whole wave), shift excursions, odd
    characters inside them: tests/golden/consensus_long_segments.json.gz, from the unmodified reference function."""
    from sniffles_amd import consensus
    probs = gu.load("consensus_long_segments")["problems"]
    got = consensus.novel_from_reads_batch([(p["best"], p["others"], p["skip"]) for p in probs], klen=6)
    assert [i for i, (g, p) in enumerate(zip(got, probs)) if g != p["expected"]] == []
    assert sum(p["expected"] != p["best"] for p in probs) >= 20


@pytest.mark.gpu
def test_novel_from_reads_signature_and_errors():
    from sniffles_amd import consensus, lib

    class Lead:
        def __init__(self, seq):
            self.seq = seq

    p = load()[3]
    out = consensus.novel_from_reads(Lead(p["best"]), [Lead(o) for o in p["others"]], klen=p["klen"], skip=p["skip"],
                                     skip_repetitive=p["skip"])
    assert out == p["expected"]
    with pytest.raises(lib.SnifflesAmdError):      # range(0, n, 0) raises in the reference as well
        consensus.novel_from_reads(Lead("ACGTACGTACGT"), [Lead("ACGTACGTACGT")], klen=6, skip=0, skip_repetitive=0)
    with pytest.raises(lib.SnifflesAmdError):      # a k-mer is one 64-bit word
        consensus.novel_from_reads(Lead("ACGTACGTACGT" * 3), [Lead("ACGTACGTACGT" * 3)], klen=9, skip=3, skip_repetitive=3)


# ---------------------------------------------------------------------------------------------------------------------------
# What the reference accepts beyond its own call site (consensus.py:280-310): skip_repetitive != skip, the byte '-' in a
# sequence (its gap symbol - a read base '-' IS a gap there), problems beyond the workgroup kernels' limits.  Compared with the
# UNMODIFIED reference function itself (checkout, or the staged build oracle/_ref), problem by problem.
def _generic_problems(seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for k in range(70):
        L = int(rng.integers(30, 700)) if k % 9 else int(rng.integers(1800, 4200))
        truth = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)

        def noisy(err, dash):
            o = truth.copy()
            hit = rng.random(L) < err
            o[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(hit.sum()))
            if dash:
                d = rng.random(L) < dash
                o[d] = ord("-")
            if rng.random() < 0.4:          # an indel: the anchors leave the diagonal for a while
                c = int(rng.integers(0, L))
                o = np.concatenate([o[:c], rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 9))), o[c:]]) if rng.random() < 0.5 \
                    else np.concatenate([o[:c], o[min(L, c + int(rng.integers(1, 9))):]])
            return o.tobytes().decode()
        kind = k % 4     # 0: skip_repetitive != skip; 1: '-' in the other reads; 2: '-' in the best read too; 3: both
        dash_o = 0.02 if kind in (1, 2, 3) else 0.0
        dash_b = 0.01 if kind in (2, 3) else 0.0
        skip = int(rng.integers(1, 8))
        skip_rep = skip if kind in (1, 2) else int(rng.integers(1, 8))
        best = noisy(0.04, dash_b)
        others = [noisy(float(rng.choice([0.02, 0.06, 0.3])), dash_o) for _ in range(int(rng.integers(0, 14)))]
        out.append(dict(best=best, others=others, klen=int(rng.integers(4, 9)), skip=skip, skip_rep=skip_rep))
    # beyond the limits of the workgroup kernels without any oddity: more than 500 sampled positions
    truth = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2600)
    for _ in range(2):
        reads = []
        for _ in range(7):
            o = truth.copy(); hit = rng.random(2600) < 0.05; o[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(hit.sum()))
            reads.append(o.tobytes().decode())
        out.append(dict(best=reads[0], others=reads[1:], klen=6, skip=3, skip_rep=3))
    return out


def _check_generic(monkeypatch, load):
    import ref_harness as rh
    from sniffles_amd import consensus
    if load is not None:
        monkeypatch.setattr(consensus._lib, "load", load)
    ref = rh.load_reference()

    class Lead:
        def __init__(self, seq):
            self.seq = seq
    probs = _generic_problems(17)
    exp = [ref.consensus.novel_from_reads(Lead(p["best"]), [Lead(o) for o in p["others"]], p["klen"], p["skip"], p["skip_rep"]) for p in probs]
    assert sum(e != p["best"] for e, p in zip(exp, probs)) >= 15
    by_klen = {}
    for i, p in enumerate(probs):
        by_klen.setdefault(p["klen"], []).append(i)
    for klen, idx in by_klen.items():       # one launch per k-mer length, mixed with problems the workgroup kernels take
        got = consensus.novel_from_reads_batch([(probs[i]["best"], probs[i]["others"], probs[i]["skip"], probs[i]["skip_rep"]) for i in idx], klen=klen)
        assert [i for i, g in zip(idx, got) if g != exp[i]] == []
    p = probs[0]
    assert consensus.novel_from_reads(Lead(p["best"]), [Lead(o) for o in p["others"]], p["klen"], p["skip"], p["skip_rep"]) == exp[0]


needs_ref = pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")


@needs_ref
def test_gap_bytes_other_anchor_steps_and_oversize_problems_match_the_reference_emu(monkeypatch):
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    _check_generic(monkeypatch, E.lib)


@needs_ref
@pytest.mark.gpu
def test_gap_bytes_other_anchor_steps_and_oversize_problems_match_the_reference_gpu(monkeypatch):
    _check_generic(monkeypatch, None)


# The workgroup kernels over the sampling steps they can meet (skip == skip_repetitive, no gap bytes): skip 1-2 probe more than five
# positions per k-mer (the loop form), 3-7 the unrolled form, >= 8 sends even a short call to the LARGE class (a step's bytes no longer
# fit the k-mer word) and >= 16 needs the third register word; k-mer lengths 4-7; calls with more than 64 other reads.
def _wave_class_problems(seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for k in range(96):
        L = int(rng.integers(20, 385)) if k % 3 == 0 else int(rng.integers(385, 1500)) if k % 3 == 1 else int(rng.integers(1500, 7000))
        if k % 16 == 15 or k % 32 == 14:
            L = int(rng.integers(40, 300))
        truth = rng.choice(acgt, L)

        def noisy(err):
            o = truth.copy()
            hit = rng.random(L) < err
            o[hit] = rng.choice(acgt, int(hit.sum()))
            for _ in range(int(rng.integers(0, 3))):     # indels: the anchors leave the diagonal for a while
                c = int(rng.integers(0, len(o)))
                o = np.concatenate([o[:c], rng.choice(acgt, int(rng.integers(1, 12))), o[c:]]) if rng.random() < 0.5 \
                    else np.concatenate([o[:c], o[min(len(o), c + int(rng.integers(1, 12))):]])
            if rng.random() < 0.1:
                o[int(rng.integers(0, len(o)))] = ord("N")                     # a byte outside A/C/G/T: the vote's escape list
            return o.tobytes().decode()
        klen = int(rng.integers(4, 8))
        smin = max(1, -(-(L - klen) // 500))              # at most 500 sampled positions: the workgroup kernels' limit
        if L <= 384 and k % 4:
            skip = int(rng.integers(smin, 8))             # SMALL when the call has at most 120 positions, else LARGE
        else:
            skip = max(smin, int(rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 23]))) if k % 2 else smin + int(rng.integers(0, 10))
        n_others = int(rng.integers(65, 90)) if k % 16 == 15 else int(rng.integers(255, 270)) if k % 32 == 14 else int(rng.integers(0, 18))
        out.append(dict(best=noisy(0.04), others=[noisy(float(rng.choice([0.02, 0.06, 0.3]))) for _ in range(n_others)],
                        klen=klen, skip=skip, skip_rep=skip))
    return out


def _check_wave_classes(monkeypatch, load):
    import ref_harness as rh
    from sniffles_amd import consensus
    if load is not None:
        monkeypatch.setattr(consensus._lib, "load", load)
    ref = rh.load_reference()

    class Lead:
        def __init__(self, seq):
            self.seq = seq
    probs = _wave_class_problems(29)
    exp = [ref.consensus.novel_from_reads(Lead(p["best"]), [Lead(o) for o in p["others"]], p["klen"], p["skip"], p["skip_rep"]) for p in probs]
    assert sum(e != p["best"] for e, p in zip(exp, probs)) >= 25
    by_klen = {}
    for i, p in enumerate(probs):
        by_klen.setdefault(p["klen"], []).append(i)
    for klen, idx in by_klen.items():
        got = consensus.novel_from_reads_batch([(probs[i]["best"], probs[i]["others"], probs[i]["skip"], probs[i]["skip_rep"]) for i in idx], klen=klen)
        assert [(i, probs[i]["skip"], len(probs[i]["best"])) for i, g in zip(idx, got) if g != exp[i]] == []


@needs_ref
def test_workgroup_kernels_over_sampling_steps_and_kmer_lengths_match_the_reference_emu(monkeypatch):
    import emu.emu as E
    _check_wave_classes(monkeypatch, E.lib)


@needs_ref
@pytest.mark.gpu
def test_workgroup_kernels_over_sampling_steps_and_kmer_lengths_match_the_reference_gpu(monkeypatch):
    _check_wave_classes(monkeypatch, None)


def _check_pipeline_with_gap_bytes(L, oracle_mod):
    """INS sequences that hold the byte '-' inside a whole task (Lead objects built by hand can; BAM records cannot): the batch takes
    the literal thread kernels for every consensus call (View::cons_thread_only) and equals the unmodified reference - and the C
    oracle - on every field, ALT strings included."""
    import numpy as np
    import ref_harness as rh
    from sniffles_amd import lib, records, synth
    from sniffles_amd.config import SnifflesConfig
    ti = synth.gen_task(0, "chr20", 1_400_000, 30, 4)
    rng = np.random.default_rng(2)
    pool = ti.seq_pool.copy()
    pool[rng.random(len(pool)) < 0.006] = ord("-")
    ti.seq_pool = pool
    cfg = SnifflesConfig()
    cfg.qc_nm_threshold = cfg.average_regional_nm = ti.qc_nm_threshold
    kw = dict(device=0)
    with lib.Batch(cfg, [ti], **kw) as b:
        b.call_candidates(); b.finalize()
        got = records.records(b.fetch(1), [ti], "final")[0]
    exp = rh.run_reference(ti)
    # (run_reference: call_candidates(keep_qc_fails=True), finalize_candidates(keep=False) returns the passing calls only)
    by_id = {r["id"]: r for r in got}
    assert len(exp["final"]) > 30 and sum(r["svtype"] == "INS" and "-" in (r["alt"] or "") for r in exp["final"]) >= 2
    for e in exp["final"]:
        assert by_id[e["id"]] == e, (e["id"], [k for k in e if by_id[e["id"]].get(k) != e.get(k)])
    assert got == records.records(oracle_mod.run(cfg, [ti], True), [ti], "final")[0]


@needs_ref
def test_task_with_gap_bytes_in_ins_sequences_emu(oracle_mod):
    import emu.emu as E
    _check_pipeline_with_gap_bytes(E.lib(), oracle_mod)


@needs_ref
@pytest.mark.gpu
def test_task_with_gap_bytes_in_ins_sequences_gpu(oracle_mod):
    _check_pipeline_with_gap_bytes(None, oracle_mod)
