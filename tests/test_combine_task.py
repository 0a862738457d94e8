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


def run_case(name, _lib=None):
    doc = gu.load(name)
    exp = doc["expected"]
    cfg = make_cfg(doc["reference_args"], exp["n_samples"])
    readers = {s: BlocksReader(exp["contig"], exp["samples"][s]) for s in range(exp["n_samples"])}
    task = parallel.CombineTask(id=7, sv_id=0, contig=exp["contig"], start=0, end=exp["contig_len"], config=cfg, _lib=_lib)
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
