"""The Python entry points mirror the reference's (Lead / LeadProvider.record_lead / Task.call_candidates /
Task.finalize_candidates -> SVCall): same objects in, same SVCall fields out, compared with the goldens the
unmodified reference produced.  CPU tier: kernels through the host emulation; GPU tier: the real library."""
import pytest

import cases
import golden_util as gu
from sniffles_amd import leadprov, parallel
from sniffles_amd.soa import SVTYPES, SOURCES, SVLEN_NONE


def leads_of(ti):
    """TaskInput -> mirror Lead objects, exactly as a ported iter_region would create them."""
    L = ti.leads
    pool = ti.seq_pool.tobytes()
    for i in range(ti.n_leads):
        svt = SVTYPES[L["svtype"][i]]
        sl, so, svlen = int(L["seq_len"][i]), int(L["seq_off"][i]), int(L["svlen"][i])
        ld = leadprov.Lead(read_id=int(L["read_id"][i]), read_qname=ti.qname(int(L["qname_id"][i])), contig=ti.contig,
                           ref_start=int(L["ref_start"][i]), ref_end=int(L["ref_end"][i]), qry_start=int(L["qry_start"][i]),
                           qry_end=int(L["qry_end"][i]), strand="-" if L["strand"][i] else "+", mapq=int(L["mapq"][i]),
                           nm=float(L["nm"][i]), source=SOURCES[L["source"][i]], svtype=svt,
                           svlen=None if svlen == int(SVLEN_NONE) else svlen,
                           seq=None if sl < 0 else pool[so:so + sl].decode("latin-1"), hap=str(int(L["hap"][i])),
                           phase_set=ti.ps_name(int(L["ps_rank"][i])), is_sa=bool(L["is_sa"][i]), read_len=int(L["read_len"][i]))
        if svt == "BND":
            ld.bnd_info = leadprov.SVCallBNDInfo(ti.contig_name(int(L["mate_contig"][i])), int(L["mate_ref_start"][i]),
                                                 bool(L["bnd_is_first"][i]), bool(L["bnd_is_reverse"][i]))
        yield ld


def as_record(c, stage):
    gt = c.genotypes.get(0)
    rec = dict(id=c.id, contig=c.contig, pos=c.pos, end=c.end, svtype=c.svtype, svlen=c.svlen, support=c.support, qual=c.qual,
               precise=c.precise, fwd=c.fwd, rev=c.rev, filter=c.filter, qc=c.qc, nm=c.nm, alt=c.alt,
               stdev_pos=c.info.get("STDEV_POS"), stdev_len=c.info.get("STDEV_LEN"), support_long=c.info.get("SUPPORT_LONG"),
               support_sa=c.info.get("SUPPORT_SA"),
               cov=[c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream],
               rnames=sorted(c.rnames),
               bnd=None if c.bnd_info is None else [c.bnd_info.mate_contig, c.bnd_info.mate_ref_start, c.bnd_info.is_first,
                                                   c.bnd_info.is_reverse])
    if stage == "final":
        rec["gt"] = None if gt is None else [gt[0], gt[1], gt[2], gt[3], gt[4], list(gt[5])]
        rec["vaf"] = c.info.get("VAF")
        rec["phase"] = c.info.get("PHASE")
    return rec


def flat_call(c):
    d = dict(c.__dict__)
    d.pop("forward_difference_sampler")
    d["rnames"] = sorted(d["rnames"])
    return repr(d)


def run_case(name, _lib):
    build, kw, _ = cases.ALL[name]
    ti = build()
    cfg = gu.make_config(kw, ti)
    exp = gu.load(name)["expected"]
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    for ld in leads_of(ti):
        lp.record_lead(ld, int(ld.ref_start / cfg.cluster_binsize) * cfg.cluster_binsize)
    for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
        lp.record_read(s, e, hp)
    if ti.nmask_start is not None:       # _mask_N_coverage through a FASTA stand-in with 'N' on the task's mask intervals
        import numpy as np
        seq = np.full(ti.contig_len, ord("a"), np.uint8)
        for a, e in zip(ti.nmask_start.tolist(), ti.nmask_end.tolist()):
            seq[a:e] = ord("N")

        class Fasta:
            def fetch(self, contig, start=None, end=None):
                assert contig == ti.contig
                return seq[start:end].tobytes().decode("ascii")
        cfg.reference = "reference.fa"
        lp._mask_N_coverage(fasta=Fasta())
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task.lead_provider = lp
    task.tandem_repeats = None if ti.tr_start is None else list(zip(ti.tr_start.tolist(), ti.tr_end.tolist()))
    if "error" in exp:
        with pytest.raises(UnboundLocalError):
            task.call_candidates(True, cfg)
        return
    cands = task.call_candidates(True, cfg)
    assert [as_record(c, "cand") for c in cands] == exp["candidates"]
    assert task.coverage_average_total == exp["coverage_average_total"]
    assert task.sv_id == ti.sv_id_start + len(cands)
    final = task.finalize_candidates(cands, False, cfg)
    assert [as_record(c, "final") for c in final] == exp["final"]
    assert all(c.postprocess is None for c in final)
    # any selection of the candidates, in any order (the reference iterates whatever it is given, parallel.py:129-147): every call
    # carries its place in the batch.  Untouched, the elements are stand-ins that become calls - in their final state - when touched
    from sniffles_amd import sv as _sv
    task6 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task6.lead_provider, task6.tandem_repeats = lp, task.tandem_repeats
    c6 = task6.call_candidates(True, cfg)
    assert type(c6) is list and len(c6) == len(cands) and all(_sv.is_stand_in(c) and isinstance(c, _sv.SVCall) for c in c6)
    with pytest.raises(RuntimeError):              # a call of another task's list (finalized: it no longer carries its place)
        task6.finalize_candidates([final[0]], False, cfg)
    sel = c6[::-1][1:]
    f6 = task6.finalize_candidates(sel, False, cfg)
    assert len(f6) == len(sel) and all(a is b for a, b in zip(f6, sel)) and all(_sv.is_stand_in(c) for c in f6)
    assert [bool(c.qc) for c in f6] == [r["qc"] for r in exp["final"][::-1][1:]] and all(_sv.is_stand_in(c) for c in f6)   # (`qc` alone fills nothing)
    assert [as_record(c, "final") for c in f6] == exp["final"][::-1][1:]
    assert not any(_sv.is_stand_in(c) for c in f6) and all(type(c) is _sv.SVCall and c.postprocess is None for c in f6)
    assert [flat_call(c) for c in f6] == [flat_call(c) for c in final[::-1][1:]]
    # ... and touched between the two calls (a caller that filters on the candidates' fields): objects at the candidate stage, mapped back
    task7 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task7.lead_provider, task7.tandem_repeats = lp, task.tandem_repeats
    c7 = task7.call_candidates(True, cfg)
    pick = [k for k, c in enumerate(c7) if k % 2 == 0 or c.svtype == "BND"]
    assert [as_record(c7[k], "cand") for k in pick] == [exp["candidates"][k] for k in pick]
    assert [as_record(c, "final") for c in task7.finalize_candidates([c7[k] for k in pick], False, cfg)] == [exp["final"][k] for k in pick]
    import copy
    import pickle
    task8 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task8.lead_provider, task8.tandem_repeats = lp, task.tandem_repeats
    c8 = task8.finalize_candidates(task8.call_candidates(True, cfg), False, cfg)
    if c8:
        # a stand-in pickles, copies, compares and prints as the call it stands for; its `__dict__` is the call's
        k8 = len(c8) // 2
        assert flat_call(pickle.loads(pickle.dumps(c8[k8]))) == flat_call(final[k8]) and type(c8[k8]) is _sv.SVCall
        assert list(c8[0].__dict__) == list(final[0].__dict__)
        if len(c8) > 2:
            assert flat_call(copy.deepcopy(c8[-1])) == flat_call(final[-1])
    # the stand-ins' source reads the batch's own result block: before the block is handed on (close, the task's next call) whatever
    # is still held becomes a call, the rest is garbage - and nothing is copied for the calls nobody kept
    task9 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task9.lead_provider, task9.tandem_repeats = lp, task.tandem_repeats
    f9 = task9.finalize_candidates(task9.call_candidates(True, cfg), False, cfg)
    if len(f9) >= 2:
        held, k9 = f9[-1], len(f9) - 1
        assert _sv.is_stand_in(held)
        del f9
        task9.close()
        assert not _sv.is_stand_in(held) and type(held) is _sv.SVCall and flat_call(held) == flat_call(final[k9])
    for t_ in (task6, task7, task8, task9):
        t_.close()
    # CallTask.execute's tail in one step (filter + sort on the device, only the kept calls become objects): the same objects
    task2 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task2.lead_provider, task2.tandem_repeats = lp, task.tandem_repeats
    kept = task2.execute_calls(cfg)
    want = final if cfg.no_qc else [c for c in final if c.qc]
    want = sorted(want, key=lambda c: c.pos) if cfg.sort else want

    def flat(c):
        d = dict(c.__dict__)
        d.pop("forward_difference_sampler")
        d["rnames"] = sorted(d["rnames"])
        return repr(d)
    assert [flat(c) for c in kept] == [flat(c) for c in want]
    assert task2.sv_id == task.sv_id and task2.coverage_average_total == task.coverage_average_total
    # the same two shapes with the device work started ahead of the call (Task.prepare: a worker loop's two-deep pipeline): same objects
    task3 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task4 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    for t_ in (task3, task4):
        t_.lead_provider, t_.tandem_repeats = lp, task.tandem_repeats
    task3.prepare(cfg); task4.prepare(cfg, execute=True)
    cands3 = task3.call_candidates(True, cfg)
    assert [as_record(c, "cand") for c in cands3] == exp["candidates"]
    assert [as_record(c, "final") for c in task3.finalize_candidates(cands3, False, cfg)] == exp["final"]
    assert [flat(c) for c in task4.execute_calls(cfg)] == [flat(c) for c in want]
    assert task4.sv_id == task.sv_id
    task5 = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task5.lead_provider, task5.tandem_repeats = lp, task.tandem_repeats
    task5.prepare(cfg, execute=True)               # prepared for another call than the one that comes: redone, not reused
    assert [as_record(c, "cand") for c in task5.call_candidates(True, cfg)] == exp["candidates"]
    for t_ in (task, task2, task3, task4, task5):
        t_.close()


NAMES = ["bnd_first_error", "bnd_stale_end", "merge_inner", "long_ins", "phase_rescue", "consensus_quirks",
         "chr21_30x_mosaic", "fuzz_4_2", "single_leads_noqc", "nmask_cov"]


@pytest.mark.parametrize("name", NAMES)
def test_task_entry_points_emulated(name):
    import emu.emu as E
    run_case(name, E.lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES + ["chr20_30x_ont", "chr22_60x_hifi"])
def test_task_entry_points_gpu(name):
    run_case(name, None)


def test_bulk_materialisation_equals_record_by_record():
    """sv.materialize_candidates / sv.apply_final build the same objects as fill_candidate / fill_final per record."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd import lib, sv
    for name in ("fuzz_4_2", "bnd_stale_end", "chr21_30x_mosaic", "single_leads_noqc"):
        build, kw, _ = cases.ALL[name]
        ti = build()
        cfg = gu.make_config(kw, ti)
        with lib.Batch(cfg, [ti], device=0) as b:
            b.call_candidates()
            r0 = b.fetch(0)
            n = len(r0.calls)
            one = [sv.fill_candidate(sv.new_call(), r0, i, ti) for i in range(n)]
            bulk = sv.materialize_candidates(r0, ti, 0, n)
            strip = lambda c: repr({k: v for k, v in c.__dict__.items() if k != "forward_difference_sampler"})  # noqa: E731
            assert [strip(c) for c in one] == [strip(c) for c in bulk]
            b.finalize()
            r1 = b.fetch(1)
            for i, c in enumerate(one):
                sv.fill_final(c, r1, i, ti)
            sv.apply_final(bulk, r1, ti)
            for a, c in zip(one, bulk):
                da, dc = dict(a.__dict__), dict(c.__dict__)
                fa, fc = da.pop("forward_difference_sampler"), dc.pop("forward_difference_sampler")
                assert repr(da) == repr(dc) and fa.__dict__ == fc.__dict__
                assert [(k, type(v)) for k, v in da.items()] == [(k, type(v)) for k, v in dc.items()]


def test_c_materialiser_equals_the_python_twin(oracle_mod):
    """sniffles_amd._snf_fast (csrc/snf_pyfast.c) builds the same SVCall objects as the pure-Python materialiser: every
    attribute, dict order included, for candidates and after the finalize fields, BND / INS / phased calls alike."""
    from sniffles_amd import sv
    assert sv._load_fast() is not None, "python -m sniffles_amd.build builds the extension"
    for name in ("chr20_30x_ont", "phase_rescue", "bnd_stale_end", "fuzz_3_3", "fuzz_5_4", "single_leads_noqc", "phase_off_symbolic"):
        build, kw, _ = cases.ALL[name]
        ti = build()
        cfg = gu.make_config(kw, ti)
        res = oracle_mod.run(cfg, [ti], True)
        if int(res.task_status[0]) != 0:
            continue
        n = len(res.calls)
        a = sv.materialize_candidates(res, ti, 0, n, post_cls=sv.SVCallPostprocessingInfo, batch="B")
        b = sv.materialize_candidates_py(res, ti, 0, n)
        for i, c in enumerate(b):
            c.postprocess = sv.SVCallPostprocessingInfo(batch="B", index=i)
        sv.apply_final(a, res, ti)
        sv.apply_final_py(b, res, ti)

        def flat(c):
            d = dict(c.__dict__)
            f = d.pop("forward_difference_sampler")
            return list(d.items()), list(d["info"].items()), f.__dict__
        assert len(a) == len(b) == n
        for x, y in zip(a, b):
            assert type(x) is type(y) and flat(x) == flat(y), (name, y.id)
            assert [type(v) for v in x.__dict__.values()] == [type(v) for v in y.__dict__.values()], (name, y.id)
            assert [type(v) for v in x.info.values()] == [type(v) for v in y.info.values()]


def test_one_walk_lead_columns_equal_the_python_passes():
    """`_snf_fast.lead_columns` (one walk over the Lead objects in C) fills the same TaskInput as the per-field Python passes:
    every column, the interned name tables in Python string order, the sequence pool."""
    import numpy as np
    from sniffles_amd import sv
    assert sv._load_fast() is not None and hasattr(sv._load_fast(), "lead_columns")
    for name in ("bnd_stale_end", "phase_rescue", "consensus_quirks", "fuzz_4_2", "long_ins", "single_leads_noqc", "chr21_30x_mosaic"):
        build, kw, _ = cases.ALL[name]
        ti = build()
        cfg = gu.make_config(kw, ti)
        out = []
        # the three forms of the input side: columns written when a lead is recorded (`_snf_fast.LeadSink`, the default), one walk over
        # the kept objects in `to_task_input` (`lead_columns`), the per-field Python passes
        for at_record_time, force_py in ((True, False), (False, False), (False, True)):
            lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len, record_time_columns=at_record_time)
            assert (lp._sink is not None) == at_record_time
            lp._force_py = force_py
            for k, ld in enumerate(leads_of(ti)):
                if k % 2:
                    lp.record_lead(ld, 0)
                else:
                    lp.record_lead(ld)            # (pos_leadtab is optional, as in the mirror's signature)
            for s, e, hp in zip(ti.read_start.tolist(), ti.read_end.tolist(), ti.read_hp.tolist()):
                lp.record_read(s, e, hp)
            out.append(lp.to_task_input(ti.task_id, ti.sv_id_start, None, ti.qc_nm_threshold))
        b = out[-1]
        for a in out[:-1]:
            assert a.n_leads == b.n_leads == ti.n_leads
            for f in a.leads:
                x, y = a.leads[f], b.leads[f]
                assert x.dtype == y.dtype and (np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)), (name, f)
            assert a.qnames == b.qnames and a.ps_names == b.ps_names and a.contig_names == b.contig_names
            assert np.array_equal(a.seq_pool, b.seq_pool)
            assert np.array_equal(a.read_start, b.read_start) and np.array_equal(a.read_end, b.read_end) and np.array_equal(a.read_hp, b.read_hp)


def test_record_time_columns_are_a_snapshot_and_reject_bad_leads():
    """`record_lead` reads the lead when it is recorded: a later change of the object does not reach the task input (the reference's
    callers record finished leads); a lead whose attributes cannot be read raises there and leaves no row; name ranks follow Python
    string order (code points) also beyond ASCII."""
    import numpy as np
    from sniffles_amd import sv
    fast = sv._load_fast()
    assert hasattr(fast, "LeadSink") and hasattr(fast, "rank_strings")
    build, kw, _ = cases.ALL["fuzz_4_2"]
    ti = build()
    cfg = gu.make_config(kw, ti)
    lp = leadprov.LeadProvider(cfg, 0, ti.contig, contig_len=ti.contig_len)
    leads = list(leads_of(ti))
    for ld in leads:
        lp.record_lead(ld, 0)
    before = lp.to_task_input(ti.task_id, ti.sv_id_start, None, ti.qc_nm_threshold)
    bad = leadprov.Lead(**{**leads[0].__dict__, "svtype": "NOT_A_TYPE"})
    with pytest.raises(KeyError):
        lp.record_lead(bad, 0)
    leads[0].ref_start += 1000                      # after the fact: not seen
    after = lp.to_task_input(ti.task_id, ti.sv_id_start, None, ti.qc_nm_threshold)
    assert after.n_leads == before.n_leads == len(leads)
    for f in before.leads:
        assert np.array_equal(before.leads[f], after.leads[f], equal_nan=before.leads[f].dtype.kind == "f")
    assert np.array_equal(before.seq_pool, after.seq_pool)
    names = ["read\u00e9", "read", "Read", "read10", "read2", "\U0001f600", "z" * 40, "z" * 39 + "y", "", "readZ", "\u4e2d"]
    rng = np.random.default_rng(5)
    many = sorted({"".join(chr(int(c)) for c in rng.choice([48, 49, 65, 97, 233, 0x4e2d, 0x1f600], int(rng.integers(0, 14)))) for _ in range(4000)})
    for pool in (names, [many[j] for j in rng.permutation(len(many))]):       # (more than 64 names: the radix path)
        in_order, rank = fast.rank_strings(pool)
        rank = np.frombuffer(rank, np.int64)
        assert in_order == sorted(pool) and [int(r) for r in rank] == [in_order.index(x) for x in pool]
