"""The drop-in seams IN SITU: `sniffles.parallel.Task.call_candidates` /
`finalize_candidates` of the UNMODIFIED, imported reference are replaced by the bodies of `sniffles_amd/parallel.py` with
`svcall_cls=sniffles.sv.SVCall`, the reference's own `LeadProvider` feeds them (`record_lead` + the reads `iter_region`
accepted), and the reference's own `CallTask.execute` (`/root/reference/src/sniffles/parallel.py:255-297`) runs around the
library.  Every `CallResult` is pickled through a `multiprocessing` pipe as the worker protocol does (`parallel.py:757`),
then the reference's VCF writer writes it; the text must equal the unpatched reference's on the same BAM.
Needs the reference: its checkout (build container) or the byte-compiled staged build `oracle/_ref` that `oracle/make_ref.py` makes during `build()` and that travels to the GPU box - the GPU twin runs there."""
import os
import pickle

import pytest

import cases

pytestmark = pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")

FIXED = dict(command="sniffles --input sample.bam --vcf out.vcf", start_date="2026/01/01 00:00:00")
NAMES = ["sample_two_contigs_12x", "sample_mosaic_20x", "sample_tandem_repeats_15x", "sample_splits_14x", "sample_noqc_10x"]


def run_both(name, emu_lib, **kw):
    import ref_harness
    import ref_insitu
    build, args = {**cases.SAMPLES, **cases.SAMPLES_EMU}[name]
    recs = build()
    plain = ref_insitu.run_call_sample(recs, args, FIXED, via=lambda r: r)
    with ref_insitu.patched(emu_lib) as ref:
        got = ref_insitu.run_call_sample(recs, args, FIXED, **kw)
        # (inside the context: the classes are the reference's own either way)
        SVCall, BND = ref.sv.SVCall, ref.sv.SVCallBNDInfo
        seen = dict(ref.insitu_seen)
    if not kw.get("in_child"):       # (a forked worker counts in its own address space)
        n_tasks = len(got["results"])
        assert seen["call_candidates"] == n_tasks and seen["finalize_candidates"] == n_tasks, seen
        assert seen["reads"] == got["read_count"] and seen["leads"] > 0 and seen["calls"] > 0, seen
    assert ref.parallel.Task.call_candidates.__module__ == "sniffles.parallel"      # patches gone again
    return plain, got, SVCall, BND, ref_harness


def check(plain, got, SVCall, BND):
    assert got["read_count"] == plain["read_count"]
    n = 0
    for res, ref_res in zip(got["results"], plain["results"]):
        assert type(res).__module__ == "sniffles.result" and type(res).__name__ == "CallResult"
        assert res.svcount == ref_res.svcount and res.coverage_average_total == ref_res.coverage_average_total
        for c, r in zip(res.svcalls, ref_res.svcalls):
            assert type(c) is SVCall                               # the reference's own class, not a mirror
            assert c.bnd_info is None or type(c.bnd_info) is BND
            assert c.postprocess is None                            # SVCall.finalize() ran (parallel.py:199)
            again = pickle.loads(pickle.dumps(c))
            assert type(again) is SVCall and again.__dict__.keys() == c.__dict__.keys()
            assert (c.pos, c.svtype, c.svlen, c.support, c.genotypes, c.alt, c.id, c.filter) == \
                   (r.pos, r.svtype, r.svlen, r.support, r.genotypes, r.alt, r.id, r.filter)
            n += 1
    assert n > 0
    assert got["vcf"] == plain["vcf"]                               # the reference writer's text, character by character


@pytest.mark.parametrize("name", NAMES)
def test_reference_execute_around_the_library_emu(name):
    import emu.emu as E
    plain, got, SVCall, BND, _ = run_both(name, E.lib())
    check(plain, got, SVCall, BND)


def test_reference_worker_process_sends_the_result_to_the_parent_emu():
    """The reference's process layout: the task executes in a forked worker, the parent receives the pickled CallResult."""
    import emu.emu as E
    plain, got, SVCall, BND, _ = run_both("sample_two_contigs_12x", E.lib(), in_child=True)
    check(plain, got, SVCall, BND)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)          # all of them, `sample_noqc_10x` (the --no-qc sort path) included
def test_reference_execute_around_the_library_gpu(name):
    plain, got, SVCall, BND, _ = run_both(name, None)
    check(plain, got, SVCall, BND)
