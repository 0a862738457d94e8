"""The SNF container (sniffles_amd/snf.py) against real `.snf` files written by the UNMODIFIED reference
(tests/golden/snf_*_s*.snf, oracle/make_golden.py::main_snf): reading them, merging over them, and writing files whose
blocks, candidates and downsampled coverages (GPU kernel `snf_batch_block_coverage`) equal the reference's.
CPU tier: kernels through the host emulation; GPU tier: the real library."""
import json
import os
import pickletools

import numpy as np
import pytest

import cases
import golden_util as gu
import snf_util as su
from sniffles_amd import leadprov, parallel, snf, sv
from sniffles_amd.config import SnifflesConfig
from test_combine import group_record, make_cfg
from test_dropin_api import leads_of

NAME = "snf_3samples_lowcov"
COMBINED = "combine_task_3samples_lowcov"


def golden_path(doc, s):
    return os.path.join(gu.GOLDEN_DIR, doc["files"][s]["file"])


def norm(rec):
    """JSON round trip (tuples -> lists, int keys -> str) so that fresh records compare with the stored ones."""
    return json.loads(json.dumps(rec, sort_keys=True))


def test_fixture_files_are_the_generated_ones():
    doc = gu.load(NAME)
    for s, f in enumerate(doc["files"]):
        assert su.sha(golden_path(doc, s)) == f["sha256"]


def test_reads_reference_files():
    doc = gu.load(NAME)
    tis = cases.SNF_FILES[NAME][0]()
    for s, ti in enumerate(tis):
        f = snf.SNFile.open(golden_path(doc, s), SnifflesConfig())
        assert not f.reqc          # written by build 2.8.1-dev
        got = norm(su.file_record(f, ti.contig, sv.TYPES))
        f.close()
        assert got == doc["files"][s]["record"]
        assert f.read_blocks("chrNone", 0) is None and f.read_blocks(ti.contig, 10 ** 9) is None
    # sample 1 was written with --output-rnames: the supporting read names travel through the container
    r1 = doc["files"][1]["record"]["blocks"]
    assert any(c["rnames"] for parts in r1.values() for p in parts for v in p["cands"].values() for c in v)


def run_combine_over_files(_lib):
    doc = gu.load(NAME)
    exp = gu.load(COMBINED)
    want = exp["expected"]
    cfg = make_cfg(exp["reference_args"], want["n_samples"])
    readers = {s: snf.SNFile.open(golden_path(doc, s), cfg) for s in range(want["n_samples"])}
    task = parallel.CombineTask(id=7, sv_id=0, contig=want["contig"], start=0, end=want["contig_len"], config=cfg)
    got = [group_record(c) for c in task.execute(readers)]
    for r in readers.values():
        r.close()
    assert len(got) == len(want["calls"])
    for g, w in zip(got, want["calls"]):
        # sample 1's file carries read names (--output-rnames); the combined records do not compare them
        assert gu.diff_records([g], [w]) == [], (g["id"], w["id"])


def test_combine_over_reference_files_emu():
    import emu.emu as E
    run_combine_over_files(E.lib())


@pytest.mark.gpu
def test_combine_over_reference_files_gpu():
    run_combine_over_files(None)


def write_sample(ti, path, cfg, _lib):
    """One sample through this package: the task's candidates -> part file -> final .snf (CallTask.execute's SNF tail
    and the main program's write_results)."""
    cfg.qc_nm_threshold = cfg.average_regional_nm = ti.qc_nm_threshold
    cfg.contig_lengths = [(ti.contig, int(ti.contig_len))]
    cfg.snf = path
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    for ld in leads_of(ti):
        lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_read(s, e, hp)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg,
                             lead_provider=lp)
    task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    cands = task.call_candidates(False, cfg)
    task.finalize_candidates(cands, True, cfg)
    part = task.write_snf_part(cands, f"{path}.tmp_{task.id}.snf")
    task.close()
    with pytest.raises(RuntimeError, match="device batch"):
        snf.SNFile(cfg, False).annotate_block_coverages(lp)       # no CPU fallback once the batch is gone
    out = snf.SNFile(cfg, open(path, "wb"))
    out.add_result(part)
    n = out.write_results(cfg, [ti.contig])
    out.close()
    assert not os.path.exists(part.snf_filename)
    return n


def run_write(tmp_path, _lib):
    doc = gu.load(NAME)
    tis = cases.SNF_FILES[NAME][0]()
    for s, ti in enumerate(tis):
        assert gu.input_sha(ti) == doc["input_sha"][s]
        want = doc["files"][s]["record"]
        cfg = SnifflesConfig(output_rnames="--output-rnames" in doc["files"][s]["args"])
        path = str(tmp_path / f"s{s}.snf")
        n = write_sample(ti, path, cfg, _lib)
        assert n == want["snf_candidate_count"]
        f = snf.SNFile.open(path, cfg)
        got = norm(su.file_record(f, ti.contig, sv.TYPES))
        f.close()
        assert sorted(got["blocks"]) == sorted(want["blocks"])
        for b in want["blocks"]:
            assert got["blocks"][b][0]["coverage"] == want["blocks"][b][0]["coverage"], (s, b)
            for t in sv.TYPES:
                assert gu.diff_records(got["blocks"][b][0]["cands"][t], want["blocks"][b][0]["cands"][t]) == [], (s, b, t)
        assert got == want
        yield path, ti


def test_writes_what_the_reference_writes_emu(tmp_path):
    import emu.emu as E
    for path, ti in run_write(tmp_path, E.lib()):
        # every class in the pickles is named as the reference names it: the files load in the reference
        f = snf.SNFile.open(path, SnifflesConfig())
        mods, seen_ref = set(), False
        import gzip
        with open(path, "rb") as h:
            raw = h.read()
        for b, parts in f.index[ti.contig].items():
            for start, length in parts:
                data = gzip.decompress(raw[f.header_length + start:f.header_length + start + length])
                # the only non-builtin classes named in the stream are the reference's
                assert b"sniffles_amd" not in data
                seen_ref = seen_ref or b"sniffles.sv" in data     # (a block of single-break candidates only is empty)
                mods.update(arg.split(" ")[0] for op, arg, _ in pickletools.genops(data) if op.name == "GLOBAL")
        f.close()
        assert mods <= {"sniffles.sv"} and seen_ref
    import sys
    assert "sniffles.sv" not in sys.modules or getattr(sys.modules["sniffles.sv"], "__file__", None)   # stand-in removed


@pytest.mark.gpu
def test_writes_what_the_reference_writes_gpu(tmp_path):
    assert len(list(run_write(tmp_path, None))) == 3


def test_reference_reads_our_files(tmp_path):
    """In the build container the unmodified reference opens the files this package writes and finds its own classes
    with the same content (the GPU box has no /root/reference: skipped there)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(gu.GOLDEN_DIR), "..", "oracle"))
    import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference sources not present")
    import emu.emu as E
    doc = gu.load(NAME)
    ref = rh.load_reference()
    for s, (path, ti) in enumerate(run_write(tmp_path, E.lib())):
        f = rh.open_reference_snf(path)
        blocks = f.read_blocks(ti.contig, int(sorted(f.index[ti.contig], key=int)[0]))
        objs = [c for t in ref.sv.TYPES for c in blocks[0][t]]
        assert objs and all(type(c) is ref.sv.SVCall for c in objs)
        assert norm(su.file_record(f, ti.contig, ref.sv.TYPES)) == doc["files"][s]["record"]
        f.close()


def dense_block_coverage(ti, binsize):
    cov = np.zeros(ti.contig_len, np.uint16)
    for s, e in zip(ti.read_start.tolist(), ti.read_end.tolist()):
        cov[s:e] += 1                                   # (uint16: wraps at 65536 like the reference's vector, leadprov.py:451, 510)
    if getattr(ti, "nmask_start", None) is not None:    # _mask_N_coverage (leadprov.py:420-443)
        for a, e in zip(ti.nmask_start.tolist(), ti.nmask_end.tolist()):
            cov[a:e] = 0
    pad = -len(cov) % binsize
    return [round(x) for x in np.pad(cov, (0, pad), mode="constant").reshape(-1, binsize).mean(axis=1)]


def check_block_coverage(tis, _lib, binsizes):
    """`snf_batch_block_coverage` against the reference's formula on the dense vector (snf.py:257-258), for every task of
    a batch, odd bin sizes (ties of the half-to-even rounding), sub-ranges and bins beyond the padded vector."""
    from sniffles_amd import lib
    with lib.Batch(SnifflesConfig(), tis, device=0) as b:
        early = b.block_coverage(0, 500, 0, 4)      # the read index exists from the upload on
        b.call_candidates()
        assert b.block_coverage(0, 500, 0, 4).tolist() == early.tolist()
        for k, ti in enumerate(tis):
            for bs in binsizes:
                want = dense_block_coverage(ti, bs)
                got = b.block_coverage(k, bs, 0, len(want) + 2)
                assert got[:len(want)].tolist() == want, (k, bs)
                assert got[len(want):].tolist() == [-1, -1]
                lo = len(want) // 3
                assert b.block_coverage(k, bs, lo, 11).tolist() == (want + [-1] * 11)[lo:lo + 11]
        with pytest.raises(lib.SnifflesAmdError):
            b.block_coverage(len(tis), 500, 0, 4)
        assert b.block_coverage(0, 500, 0, 0).shape == (0,)


def test_block_coverage_kernel_emu():
    import emu.emu as E
    from sniffles_amd import synth
    tis = [synth.gen_fuzz(7, task_id=0), synth.gen_task(1, "chr20", 777_777, 30, 3), cases.SNF_FILES[NAME][0]()[0]]
    check_block_coverage(tis, E.lib(), (500, 7, 2, 1000))


def masked_and_deep_tasks():
    a, b = cases.case_nmask_cov(), cases.case_deep_wrap_cov()
    b.task_id = 1
    return [a, b]


def test_block_coverage_of_masked_and_wrapping_vectors_emu():
    """Reference 'N' mask and the uint16 wrap at depth >= 65536 in the per-bin means of the SNF writer (snf.py:249-267 on the
    vector of leadprov.py:420-451)."""
    import emu.emu as E
    check_block_coverage(masked_and_deep_tasks(), E.lib(), (500, 7, 100))


def test_block_coverage_of_masked_and_wrapping_vectors_simt():
    from emu import simt as S
    check_block_coverage(masked_and_deep_tasks(), S.lib(), (500, 100))


@pytest.mark.gpu
def test_block_coverage_of_masked_and_wrapping_vectors_gpu():
    check_block_coverage(masked_and_deep_tasks(), None, (500, 7, 100))


def test_exact_coverage_walk_equals_the_closed_forms(monkeypatch, oracle_mod):
    """SNF_COV_EXACT=1 sends every task through the walk over read starts / ends that masked or wrapping tasks take
    (snf_cov.h): same coverage.mean(), same block coverages, same calls as the closed forms on ordinary data."""
    import emu.emu as E
    from sniffles_amd import lib, records, synth
    E.lib()                                              # the host tier becomes the library of this test
    tis = [synth.gen_fuzz(7, task_id=0), synth.gen_task(1, "chr20", 777_777, 30, 3)]
    cfg = SnifflesConfig()
    exp = oracle_mod.run(cfg, tis, True)
    monkeypatch.setenv("SNF_COV_EXACT", "1")
    with lib.Batch(cfg, tis) as b:
        b.call_candidates(); b.finalize()
        got = b.fetch(1)
    for t in range(2):
        assert records.diff_results(got, t, exp, t) == []
    check_block_coverage(tis, E.lib(), (500, 7))


@pytest.mark.gpu
def test_block_coverage_kernel_gpu():
    from sniffles_amd import synth
    tis = [synth.gen_fuzz(7, task_id=0), synth.gen_task(1, "chr20", 5_000_000, 60, 3), synth.gen_task(2, "chr21", 3_000_001, 30, 4)]
    check_block_coverage(tis, None, (500, 7, 2, 1000))


def test_snf_blocks_cannot_name_arbitrary_globals():
    """An .snf block is a pickle: the reader resolves the Sniffles record classes and a fixed list of harmless globals and
    refuses everything else (the stock unpickler would import and call whatever the file names)."""
    import pickle
    import pytest as _pt
    from sniffles_amd import sv
    f = snf.SNFile(SnifflesConfig(), False)
    evil = pickle.dumps({"INS": [], "x": __import__("os").getcwd})          # names posix.getcwd
    with _pt.raises(pickle.UnpicklingError, match="refusing"):
        f.unserialize_block(evil)
    c = sv.new_call()
    c.svtype, c.pos = "INS", 5
    f.store(c)
    back = f.unserialize_block(f.serialize_block(0))
    assert type(back["INS"][0]) is sv.SVCall and back["INS"][0].pos == 5
    assert sv.SVCall.__module__ == "sniffles_amd.sv"                          # dumping leaves the record classes alone
