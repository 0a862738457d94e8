"""Seam B3: `sniffles_amd.cluster.resolve(svtype, lead_provider, config, tr)` and the `--dev-dump-clusters`
BED text against the UNMODIFIED reference's own `cluster.resolve` (cluster.py:219-353) on the same tasks
(tests/golden/clusters_resolve.json.gz, oracle/ref_harness.py::run_reference_clusters): ids, bounds, repeat flags, the
number of long leads, and every cluster's leads in the reference's list order, SV type by SV type - after the merge scan
(the dump point) and as yielded (after merge_inner / resplit / resplit_bnd)."""
import pytest

import cases
import golden_util as gu
from sniffles_amd import cluster, parallel, pipeline
from sniffles_amd.soa import SVTYPES

DOC = gu.load("clusters_resolve")
NAMES = sorted(DOC)


def run_case(name, _lib=None):
    build, kw, _ = cases.ALL[name]
    ti = build()
    doc = DOC[name]
    assert gu.input_sha(ti) == doc["input_sha"]
    cfg = gu.make_config(kw, ti)
    task = parallel.CallTask(id=ti.task_id, sv_id=ti.sv_id_start, contig=ti.contig, start=0, end=ti.contig_len, config=cfg)
    task.lead_provider = pipeline._Extracted(ti)
    try:
        task.call_candidates(True, cfg)
    except UnboundLocalError:
        pass        # the reference's own failure of postprocessing.coverage comes after the clusters exist
    lp = task.lead_provider
    lp.task_input = ti
    for svtype in SVTYPES:
        want = doc["clusters"][svtype]
        assert cluster.dump_clusters_bed(lp, cfg, svtype) == want["bed"], (name, svtype)
        got = [dict(id=c.id, start=c.start, end=c.end, seed=c.seed, repeat=c.repeat, leads_long=c.leads_long,
                    leads=[[ld.read_qname, ld.ref_start, ld.svlen, ld.source] for ld in c.leads])
               for c in cluster.resolve(svtype, lp, cfg, task.tandem_repeats)]
        assert len(got) == len(want["yielded"]), (name, svtype)
        for g, w in zip(got, want["yielded"]):
            assert g == w, (name, svtype, w["id"])
    task.close()


@pytest.mark.parametrize("name", NAMES)
def test_cluster_views_match_reference_emu(name):
    import emu.emu as E
    run_case(name, E.lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_cluster_views_match_reference_gpu(name):
    run_case(name)


def test_cluster_metrics_with_positions_far_apart_emu(oracle_mod):
    """compute_metrics keeps its sums in 64 bits while the sampled positions of a bin lie within 2^27 bp of the first one and
    falls back to 128-bit sums otherwise: a bin size larger than the contig puts leads 200 Mb apart into one bin."""
    import emu.emu as E
    from sniffles_amd import lib, records, synth
    from sniffles_amd.config import SnifflesConfig
    E.lib()
    ti = synth.gen_task(0, "chr1", 248_956_422, 1, 3)
    cfg = SnifflesConfig(cluster_binsize=300_000_000)
    exp = records.records(oracle_mod.run(cfg, [ti], False), [ti], "cand")
    with lib.Batch(cfg, [ti]) as b:
        b.call_candidates()
        assert records.records(b.fetch(0), [ti], "cand") == exp
    assert int(ti.leads["ref_start"].max()) - int(ti.leads["ref_start"].min()) > (1 << 27) and len(exp[0]) > 0


def check_merged_cluster_metrics_added_or_read(oracle_mod, monkeypatch):
    """The merge scan keeps the sums behind a cluster's mean / stdev and adds them under merges (snf_stage_cluster.h::merge_walk);
    SNF_MERGE_REREAD=1 makes it read the merged cluster's leads instead, as the reference does: the same clusters, candidates and
    final records either way - on deep data too, where merged clusters pass 200 leads and the sums are dropped."""
    from sniffles_amd import lib, records, synth
    from sniffles_amd.config import SnifflesConfig
    tis = [synth.gen_task(0, "chr21", 800_000, 30, 21), synth.gen_task(1, "chr22", 200_000, 150, 22), synth.gen_fuzz(9, task_id=2)]
    cfg = SnifflesConfig()
    exp = records.records(oracle_mod.run(cfg, tis, True), tis, "final")
    got = {}
    for mode in ("0", "1"):
        if mode == "1":
            monkeypatch.setenv("SNF_MERGE_REREAD", "1")
        with lib.Batch(cfg, tis) as b:
            b.run_pass()
            got[mode] = (records.records(b.fetch(1), tis, "final"), b.fetch_clusters(1))
    assert got["0"][0] == exp and got["1"][0] == exp
    ca, cb = got["0"][1], got["1"][1]
    assert sorted(ca) == sorted(cb)
    for k in ca:
        assert (ca[k] == cb[k]).all() if hasattr(ca[k], "all") else ca[k] == cb[k], k


def test_merged_cluster_metrics_added_or_read_emu(oracle_mod, monkeypatch):
    import emu.emu as E
    E.lib()
    check_merged_cluster_metrics_added_or_read(oracle_mod, monkeypatch)


@pytest.mark.gpu
def test_merged_cluster_metrics_added_or_read_gpu(oracle_mod, monkeypatch):
    check_merged_cluster_metrics_added_or_read(oracle_mod, monkeypatch)
