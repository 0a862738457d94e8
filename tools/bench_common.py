"""Shared by bench.py and its legs (tools/bench_legs.py, tools/bench_roofline.py): the workloads of BASELINE.json's configs, the
GPU-less test mode of the N > 1 plumbing."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SNF_BENCH_EMU=1 (tests only, tests/test_multi_rank.py): the N > 1 code of this file - process group, work queue, result
# export, gather - on a GPU-less box: gloo instead of RCCL, CPU tensors, the kernels through the test tier's host emulation
# (tests/emu).  Never a result: the line says so.
EMU = os.environ.get("SNF_BENCH_EMU") == "1"
DEV = "cpu" if EMU else "cuda"


def emu_lib():
    """SNF_BENCH_EMU=1: the host tier of the test suite becomes the library this process works on (emu.emu.lib() calls
    sniffles_amd.lib.use_library)."""
    if not EMU:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu.emu as E
    return E.lib()


def dev_sync(torch):
    if not EMU:
        torch.cuda.synchronize()


def set_dev(torch, local_rank):
    if not EMU:
        torch.cuda.set_device(local_rank)



# the unmodified reference (CPython) on this path, timed in the build container only (it cannot travel to the GPU box):
# tools/time_reference.py, profiles/r01_reference_cpu.json
REFERENCE_CPYTHON = dict(sig_s=34600.0, host="build container, 1 core (profiles/r01_reference_cpu.json: unmodified "
                                             "Task.call_candidates + finalize_candidates on chr21 + chr22 of configs[1])")

WORKLOADS = {
    0: dict(name="chr20-only 30x ONT HG002-shaped germline (BASELINE.json configs[0])", contigs=["chr20"], coverage=30.0,
            gen={}, cfg={}),
    1: dict(name="30x ONT HG002-shaped whole-genome germline, 24 GRCh38 contigs per replica (BASELINE.json configs[1])",
            contigs=None, coverage=30.0, gen={}, cfg={}),
    2: dict(name="60x PacBio-HiFi HG002-shaped whole-genome germline, 24 GRCh38 contigs per replica (BASELINE.json configs[2])",
            contigs=None, coverage=60.0, gen=dict(err=0.005, read_len_mean=15000.0), cfg={}),
    3: dict(name="30x ONT HG002-shaped whole genome, --mosaic low-VAF mode (BASELINE.json configs[3])",
            contigs=None, coverage=30.0, gen=dict(mosaic_frac=0.3), cfg=dict(mosaic=True)),
}


def task_specs(args, wl, rep: int, g: int, world: int) -> list:
    """[(contig index, synth.gen_task kwargs)] of one genome replica."""
    from sniffles_amd import synth
    contigs = wl["contigs"] or synth.CONTIGS
    cov = args.coverage if args.coverage is not None else wl["coverage"]
    out = []
    for ci, c in enumerate(contigs):
        L = max(200000, int(synth.GRCH38[c] * args.scale))
        out.append((ci, dict(task_id=(g * world + rep) * 24 + ci, contig=c, contig_len=L, coverage=cov,
                             seed=1 + rep + 1000 * g, **wl["gen"])))
    return out
