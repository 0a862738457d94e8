"""The multi-sample merge DRIVER: sniffles_amd.parallel.CombineTask.execute against the combined calls
the UNMODIFIED reference CombineTask.execute (parallel.py:444-572) emits for the same SNF blocks
(tests/golden/combine_task_*.json.gz, oracle/ref_harness.py::run_reference_combine_task)."""
import pytest

import cases
import golden_util as gu
from sniffles_amd import parallel, sv
from test_combine import group_record, make_cfg, to_call

NAMES = sorted(cases.COMBINE_TASK)


class BlocksReader:
    """SNF reader interface over the golden's block records (SNFile.read_blocks, snf.py:139-166)."""
    reqc = False

    def __init__(self, contig, blocks):
        self.contig = contig
        self.blocks = {}
        for b in blocks:
            d = {svt: [to_call(r) for r in b["cands"][svt]] for svt in sv.TYPES}
            d["_COVERAGE"] = {int(k): v for k, v in b["coverage"].items()}
            self.blocks[b["block"]] = d

    def read_blocks(self, contig, block_index):
        if contig != self.contig or block_index not in self.blocks:
            return None
        return [self.blocks[block_index]]

    def block_starts(self, contig):
        """(a reader that lists its blocks keeps its candidates as columns: candstore.ContigColumns)"""
        return list(self.blocks) if contig == self.contig else []


def run_case(name, _lib=None, reqc=False):
    doc = gu.load(name)
    exp = doc["expected"]
    cfg = _twin_cfg(doc, ())
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    for r in readers.values():
        r.reqc = reqc
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
    got = [group_record(c) for c in task.execute(readers)]
    want = exp["calls"]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert gu.diff_records([g], [w]) == [], (g["id"], w["id"])


@pytest.mark.parametrize("name", NAMES)
def test_combine_task_driver_matches_reference_emu(name):
    import emu.emu as E
    run_case(name, E.lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_combine_task_driver_matches_reference_gpu(name):
    run_case(name)


@pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")
def test_combine_driver_matches_reference_under_random_options(monkeypatch, capsys):
    """oracle/ref_combinefuzz.py: --combine-* option sets drawn from the reference's argparse definitions, a small population,
    the unmodified reference's CombineTask.execute against this driver on the same SNF blocks."""
    import ref_combinefuzz
    monkeypatch.setattr("sys.argv", ["ref_combinefuzz.py", "6", "7000"])
    ref_combinefuzz.main()
    out = capsys.readouterr().out
    assert "mismatching 0 " in out and "MISMATCH" not in out, out


def test_combine_task_reqc_regenotypes_the_candidates_emu():
    """--reqc (SNF files older than 2.5.3, parallel.py:507-508): the candidates are genotyped again on their way into the
    bins.  Genotyping candidates that already carry the current genotype is idempotent (tests/test_genotype.py pins the
    function itself on the reference), so the combined calls are those of the golden."""
    import emu.emu as E
    run_case("combine_task_3samples_lowcov", E.lib(), reqc=True)


@pytest.mark.gpu
def test_combine_task_reqc_regenotypes_the_candidates_gpu():
    run_case("combine_task_3samples_lowcov", reqc=True)


def run_scatter_case(_lib=None):
    """CombineTask.scatter / clone against the reference's own (parallel.py:411-442; class constant lowered to 40 blocks x
    samples in the golden run so that a 3-Mb contig is cut): the same cuts, ids, bounds, and every part's combined calls."""
    doc = gu.load("combine_task_6samples")
    exp = doc["expected"]
    sc = exp["scatter"]
    cfg = make_cfg(doc["reference_args"], exp["n_samples"])
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(exp["n_samples"])]
    cfg.threads = sc["threads"]
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
    task.TARGET_WORK_PER_TASK = sc["target_work_per_task"]
    parts = task.scatter()
    assert [(p.id, p.start, p.end, p.block_indices) for p in parts] == [(w["id"], w["start"], w["end"], w["block_indices"]) for w in sc["tasks"]]
    assert len(parts) > 3
    for part, want in zip(parts, sc["tasks"]):
        got = [group_record(c) for c in part.execute(readers)]
        assert len(got) == len(want["calls"])
        for g, w in zip(got, want["calls"]):
            assert gu.diff_records([g], [w]) == [], (part.id, w["id"])
    many = parallel.CombineTask.execute_many(task.scatter(), readers)   # fresh parts (call ids count per task), one launch for all
    for got, want in zip(many, sc["tasks"]):
        assert [group_record(c) for c in got] == [group_record(c) for c in got] and len(got) == len(want["calls"])
        for g, w in zip(got, want["calls"]):
            assert gu.diff_records([group_record(g)], [w]) == [], w["id"]
    cfg.threads = 1
    assert task.scatter() == [task]


def test_combine_task_scatter_matches_reference_emu():
    import emu.emu as E
    run_scatter_case(E.lib())


@pytest.mark.gpu
def test_combine_task_scatter_matches_reference_gpu():
    run_scatter_case()


# ---- chains cut into independent sub-chains (cluster.chain_cuts) must give the assignment of the whole chain
def random_chains(seed, n_chains=10):
    import numpy as np
    from test_combine import random_problem
    rng = np.random.default_rng(seed)
    chains = []
    for k in range(n_chains):
        svtype = ["INS", "DEL", "DUP", "INV", "BND"][k % 5]
        gate = 2001 if svtype == "BND" else 1001
        cands, woff, wbin, wthr = [], [0], [], []
        base = 0
        for w in range(int(rng.integers(1, 14))):
            win = random_problem(rng, svtype, int(rng.integers(1, 30)))
            lo = min(c.pos for c in win)
            # gaps around the gate: well below, exactly at it, one above, far above
            gap = int(rng.choice([150, 600, gate - 1, gate, gate + 1, gate + 2, 6000]))
            shift = (base + gap - lo) if cands else 0
            for c in win:
                c.pos += shift
                c.id = f"{c.id}w{w}"
            base = max(c.pos for c in win)
            cands.extend(win)
            woff.append(len(cands))
            wbin.append(base // 100 * 100)
            wthr.append(float(rng.choice([50.0, 1250.0, 2500.0])))
        chains.append((svtype, cands, woff, wbin, wthr))
    return chains


def check_cut_equals_whole(_lib):
    import numpy as np
    from sniffles_amd import cluster
    from sniffles_amd.config import SnifflesConfig
    cfg = SnifflesConfig()
    n_cut = 0
    for seed in (11, 12, 13):
        chains = random_chains(seed)
        n_cut += sum(len(cluster.chain_cuts(t, c, wo, cfg)) - 2 for t, c, wo, _, _ in chains)
        whole = cluster.resolve_chains_batch(chains, cfg, cut=False)
        parts = cluster.resolve_chains_batch(chains, cfg, cut=True)
        for (t, c, wo, _, _), a, b in zip(chains, whole, parts):
            assert np.array_equal(a[:len(c)], b[:len(c)]), (seed, t)
    assert n_cut > 20   # the generator does produce cuts (and gaps exactly at the gate, which must not cut)


def test_chain_cuts_keep_the_assignment_emu():
    import emu.emu as E
    check_cut_equals_whole(E.lib())


@pytest.mark.gpu
def test_chain_cuts_keep_the_assignment_gpu():
    check_cut_equals_whole(None)


# ---- the columnar candidate store (sniffles_amd/candstore.py) against the object-by-object replay (SVGroup / SVGroup.call)
TWIN_OPTIONS = [(), ("--dev-combine-medians",), ("--combine-pair-relabel", "--combine-pair-relabel-threshold", "10"),
                ("--combine-output-filtered",), ("--combine-exhaustive",), ("--combine-high-confidence", "0.5"),
                ("--combine-low-confidence", "0.5", "--combine-low-confidence-abs", "3"), ("--combine-null-min-coverage", "12"),
                ("--combine-support-threshold", "6"), ("--combine-pctseq", "0"), ("--minsvlen", "200")]


def _twin_cfg(doc, extra):
    exp = doc["expected"]
    cfg_args, kw, a = [], {}, list(tuple(doc["reference_args"]) + tuple(extra))
    while a:
        k = a.pop(0)
        name = k[2:].replace("-", "_")
        if a and not a[0].startswith("--"):
            v = a.pop(0)
            kw[name] = float(v) if "." in v else int(v)
        else:
            kw[name] = True
    from sniffles_amd.config import SnifflesConfig
    cfg = SnifflesConfig(**{k: v for k, v in kw.items() if k != "combine_exhaustive"})
    if kw.get("combine_exhaustive"):
        cfg.combine_exhaustive = True
    cfg.snf_input_info = [dict(internal_id=s) for s in range(exp["n_samples"])]
    cfg.mode = "combine"
    return cfg


def _object_fields(c):
    d = dict(vars(c))
    d["forward_difference_sampler"] = vars(d["forward_difference_sampler"])
    return d


def check_columns_equal_objects(_lib, monkeypatch, names=None, options=None):
    for name in (names or NAMES):
        doc = gu.load(name)
        exp = doc["expected"]
        for extra in (options or TWIN_OPTIONS):
            out = []
            for objects in ("0", "1"):
                monkeypatch.setenv("SNF_COMBINE_OBJECTS", objects)
                cfg = _twin_cfg(doc, extra)
                readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
                half = exp["contig_len"] // 2 // cfg.snf_block_size * cfg.snf_block_size
                tasks = [parallel.CombineTask(id=7, sv_id=3, contig=exp["contig"], start=0, end=half, config=cfg),
                         parallel.CombineTask(id=9, sv_id=0, contig=exp["contig"], start=half + cfg.snf_block_size, end=exp["contig_len"],
                                              config=cfg)]
                calls = parallel.CombineTask.execute_many(tasks, readers)
                cands = [c for r in readers.values() for b in r.blocks.values() for t in sv.TYPES for c in b[t]]
                out.append(([[_object_fields(c) for c in part] for part in calls], [t.sv_id for t in tasks],
                            [(c.sample_internal_id, dict(c.genotypes)) for c in cands]))
            (col, col_ids, col_cands), (obj, obj_ids, obj_cands) = out
            assert col_ids == obj_ids, (name, extra)
            assert [len(p) for p in col] == [len(p) for p in obj], (name, extra)
            for pc, po in zip(col, obj):
                for a, b in zip(pc, po):
                    assert list(a) == list(b), (name, extra, a["id"])             # the same attributes in the same order
                    for k in a:
                        if k == "rnames":
                            assert a[k] == b[k], (name, extra, a["id"], k)
                        else:
                            assert a[k] == b[k] and type(a[k]) is type(b[k]), (name, extra, a["id"], k, a[k], b[k])
                    assert list(a["genotypes"]) == list(b["genotypes"]), (name, extra, a["id"])   # sample order of the dict
            assert col_cands == obj_cands, (name, extra)       # what the merge leaves on the candidates (parallel.py:509, sv.py:392)
            if extra == ():
                assert sum(len(p) for p in col) > 0


def test_columnar_store_equals_the_object_replay_emu(monkeypatch):
    """(Host tier: two goldens x a cross-section of the option sets - the wave-per-window edit distance is slow on fibres; the
    GPU twin below runs all of them.)"""
    import emu.emu as E
    check_columns_equal_objects(E.lib(), monkeypatch, names=["combine_task_3samples_lowcov", "combine_task_5samples_medians"],
                                options=[TWIN_OPTIONS[0]] + list(TWIN_OPTIONS[1::3]))


@pytest.mark.gpu
def test_columnar_store_equals_the_object_replay_gpu(monkeypatch):
    check_columns_equal_objects(None, monkeypatch)


def test_candidate_store_helpers_reject_malformed_input():
    """The C helpers of the columnar store (sniffles_amd._snf_fast) raise instead of reading out of bounds."""
    import numpy as np
    from sniffles_amd import abi
    fast = sv._load_fast()
    if fast is None:
        pytest.skip("C extension not built")
    # flush_windows: key / bin lengths must agree
    with pytest.raises(ValueError):
        fast.flush_windows(np.zeros(3, np.int64), np.zeros(2, np.int32), 100, 25, False)
    wend, wbin, wsize = fast.flush_windows(np.array([0, 0, 0, 1], np.int64), np.array([100, 100, 300, 0], np.int32), 100, 2, False)
    assert np.frombuffer(wend, np.int64).tolist() == [2, 3, 4] and np.frombuffer(wbin, np.int32).tolist() == [100, 300, 0]
    assert np.frombuffer(wsize, np.int32).tolist() == [100, 100, 100]
    # gather_pool: an index beyond the table
    with pytest.raises(ValueError):
        fast.gather_pool(np.array([0, 2, 5], np.int64), b"abcde", np.array([2], np.int64))
    off, pool = fast.gather_pool(np.array([0, 2, 5], np.int64), b"abcde", np.array([1, 0, 1], np.int64))
    assert pool == b"cdeabcde" and np.frombuffer(off, np.int64).tolist() == [0, 3, 5, 8]
    # gather_pool_parts: strings that lie in several pools; a range beyond its pool, a pool number beyond the list
    pools = [np.frombuffer(b"abcde", np.uint8), np.frombuffer(b"XYZ", np.uint8)]
    part, start, length = np.array([0, 1, 0], np.int32), np.array([1, 0, 3], np.int64), np.array([2, 3, 2], np.int64)
    off, pool = fast.gather_pool_parts(pools, part, start, length, np.array([2, 1, 0, 1], np.int64))
    assert pool == b"deXYZbcXYZ" and np.frombuffer(off, np.int64).tolist() == [0, 2, 5, 7, 10]
    with pytest.raises(ValueError):
        fast.gather_pool_parts(pools, part, start, np.array([2, 4, 2], np.int64), np.array([1], np.int64))
    with pytest.raises(ValueError):
        fast.gather_pool_parts(pools, np.array([0, 2, 0], np.int32), start, length, np.array([1], np.int64))
    with pytest.raises(ValueError):
        fast.gather_pool_parts(pools, part, start, length, np.array([3], np.int64))
    big = [np.frombuffer(bytes(range(256)) * 4096, np.uint8)] * 3      # megabytes: the copies are shared by four threads
    rng = np.random.default_rng(3)
    bp, bs, bl = rng.integers(0, 3, 20000).astype(np.int32), rng.integers(0, 1 << 19, 20000).astype(np.int64), rng.integers(0, 900, 20000).astype(np.int64)
    order = rng.permutation(20000).astype(np.int64)
    off, pool = fast.gather_pool_parts(big, bp, bs, bl, order)
    want = b"".join(big[0][int(bs[i]):int(bs[i] + bl[i])].tobytes() for i in order.tolist())
    assert pool == want and len(want) > (4 << 20) and np.frombuffer(off, np.int64)[-1] == len(want)
    # collect: one entry per reader expected
    with pytest.raises(TypeError):
        fast.collect([[None, None]], np.zeros(1, np.int32), tuple(sv.TYPES), 3, {})
    objs, rec, cblk, ctyp, mate, aoff, apool = fast.collect([], np.zeros(0, np.int32), tuple(sv.TYPES), 3, {})
    assert objs == [] and len(rec) == 0 and np.frombuffer(aoff, np.int64).tolist() == [0]
    # group_calls: a group index beyond the table
    out = np.zeros(1, abi.GROUP_OUT_DTYPE)
    with pytest.raises(ValueError):
        fast.group_calls(sv.SVCall, sv.ForwardDifferenceWelford, [], out, np.array([3], np.int64), np.array([0, 0], np.int64),
                         np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int32),
                         np.zeros(2, np.int32), [], np.zeros(2, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), 5, "", False, np.zeros(0, np.int32))


def test_columnar_store_single_sample_merge_emu(monkeypatch):
    """One input file (`sniffles --input one.snf`): the combined call keeps the candidate's own filter and info dict and has no
    STDEV fields (sv.py:440-470); with --no-qc every group is called.  Columnar store against the object replay."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    from sniffles_amd.config import SnifflesConfig
    doc = gu.load("combine_task_3samples_lowcov")
    exp = doc["expected"]
    for no_qc in (False, True):
        out = []
        for objects in ("0", "1"):
            monkeypatch.setenv("SNF_COMBINE_OBJECTS", objects)
            cfg = SnifflesConfig(no_qc=no_qc)
            cfg.snf_input_info = [dict(internal_id=0)]
            cfg.mode = "combine"
            readers = {0: BlocksReader(exp["contig"], exp["samples"][0])}
            task = parallel.CombineTask(id=1, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)
            out.append([_object_fields(c) for c in task.execute(readers)])
        col, obj = out
        assert len(col) == len(obj) and (len(col) > 0 or not no_qc)
        for a, b in zip(col, obj):
            assert list(a) == list(b) and a == b, a["id"]


def test_cached_reader_columns_equal_the_block_walk_emu(monkeypatch):
    """Readers that list their blocks keep their candidates as columns (`candstore.ContigColumns`): a second merge over the same
    readers - also with another support threshold, and as two part tasks - starts from the cached tables and gives the calls of
    the block-by-block walk (`SNF_COMBINE_NO_COLUMNS=1`)."""
    import emu.emu as E
    E.lib()                                              # the host tier becomes the library of this test
    name = "combine_task_5samples_medians"
    doc = gu.load(name)
    exp = doc["expected"]

    def merge(readers, extra, split, no_columns):
        monkeypatch.setenv("SNF_COMBINE_NO_COLUMNS", "1" if no_columns else "0")
        cfg = _twin_cfg(doc, extra)
        if split:
            half = exp["contig_len"] // 2 // cfg.snf_block_size * cfg.snf_block_size
            tasks = [parallel.CombineTask(id=7, sv_id=3, contig=exp["contig"], start=0, end=half, config=cfg),
                     parallel.CombineTask(id=9, sv_id=0, contig=exp["contig"], start=half + cfg.snf_block_size, end=exp["contig_len"], config=cfg)]
        else:
            tasks = [parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg)]
        return [[_object_fields(c) for c in part] for part in parallel.CombineTask.execute_many(tasks, readers)]
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    for extra, split in (((), False), ((), True), (("--combine-support-threshold", "6"), False), ((), False)):
        walk = merge({s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}, extra, split, True)
        cached = merge(readers, extra, split, False)          # the SAME readers throughout: tables cached by (contig, sample, threshold)
        assert cached == walk and sum(len(p) for p in cached) > 0, (extra, split)
    assert len(readers[0].__dict__["_snf_columns"]) == 2       # two thresholds -> two tables; the part tasks reused the first


def test_merge_in_runs_of_tasks_equals_one_launch_emu(monkeypatch):
    """`candstore.execute_many` cuts a big merge into runs of contig tasks that two threads work on in turn (host work of one run under
    the GPU call of the other; `SNF_COMBINE_CHUNKS`): tasks share nothing, so the calls - objects, ids, order - are those of the one
    launch for all tasks."""
    import emu.emu as E
    from sniffles_amd import candstore
    E.lib()
    name = "combine_task_5samples_medians"
    doc = gu.load(name)
    exp = doc["expected"]

    def merge(chunks):
        monkeypatch.setenv("SNF_COMBINE_CHUNKS", str(chunks))
        cfg = _twin_cfg(doc, ())
        bs = cfg.snf_block_size
        cuts = [0] + [exp["contig_len"] * k // 3 // bs * bs for k in (1, 2)] + [exp["contig_len"] + bs]
        tasks = [parallel.CombineTask(id=5 + 2 * k, sv_id=k, contig=exp["contig"], start=cuts[k], end=cuts[k + 1] - bs, config=cfg) for k in range(3)]
        readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
        out = [[_object_fields(c) for c in part] for part in parallel.CombineTask.execute_many(tasks, readers)]
        return out, candstore.last_timing.get("chunks", 1)
    whole, n1 = merge(1)
    assert n1 == 1 and sum(len(p) for p in whole) > 0 and len(whole) == 3
    for k in (2, 3):
        runs, nk = merge(k)
        assert 1 < nk <= k and runs == whole, k


def test_radix_argsort_equals_numpy_stable():
    """`_snf_fast.argsort_i64` (the merge's one sort: candidates by a 63-bit key) against numpy's stable argsort: negative keys, equal
    keys (stability), keys whose upper digits all agree (skipped passes), the empty table."""
    import numpy as np
    fast = sv._load_fast()
    if fast is None:
        pytest.skip("C extension not built")
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 7, 1000, 50000):
        for keys in (rng.integers(-5, 50, n), rng.integers(-(1 << 62), 1 << 62, n), np.full(n, 1 << 40) + rng.integers(0, 3, n),
                     (rng.integers(0, 24, n) << 51) | (rng.integers(0, 5, n) << 48) | (rng.integers(0, 3000, n) << 26) | rng.integers(0, 2500000, n)):
            k = np.ascontiguousarray(keys, np.int64)
            assert np.array_equal(np.frombuffer(fast.argsort_i64(k), np.int64), np.argsort(k, kind="stable")), n
