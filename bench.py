#!/usr/bin/env python3
"""Benchmark of the MI355X-native Sniffles2 hot path (BASELINE.json metric).

One "step" = one full pass of the hot path (Task.call_candidates + Task.finalize_candidates of every contig task of the workload:
binning, clustering, candidate calls, coverage, QC, genotyping, phasing, INS consensus - `snf_batch_pass`) with the signature
tables already resident in HBM, INCLUDING the arrival of the result block in host memory (what `CallTask.execute` returns: the
QC-passing calls per task sorted by position, their read names and ALT bytes; `--output candidates`: every candidate record) and,
for N > 1, the gather on rank 0.

Workloads (--config, BASELINE.json `configs`; all seeded synthetic signature sets):
  0  chr20-only 30x ONT germline (the reference's CPU-runnable plumbing case)
  1  30x ONT HG002-shaped whole genome, germline           <- default, the configuration the metric is quoted on
  2  60x PacBio-HiFi-shaped whole genome (INS consensus heavy: err 0.5 %, 15-kb reads)
  3  30x ONT whole genome, --mosaic (30 % of the sites at VAF 0.05-0.2)
  4  population merge: 10 HG002-shaped samples -> combine (CombineTask.execute: candidates resident as columns -> merged VCF records)
--scaling weak (default): N genome replicas, the 24*N contig tasks sharded longest-first over the ranks.
--scaling strong: ONE genome; its contigs are partitioned into one LPT-balanced contig SET per rank and pass, which the ranks claim
  from a shared work queue (sniffles_amd.dist.TaskQueue over the process group's store); a set is one device batch; passes pipelined.
The only collective is the gather on rank 0.  On one node (the default) every rank's kernels store its result into a
shared-memory segment that rank 0 maps as well (sniffles_amd.dist.SharedLanding: N PCIe links in parallel) and only the
layouts are gathered; SNF_BENCH_GATHER=rccl gathers the result blocks over RCCL (dist.gather_results).

At N = 1 the line also carries
  wall_clock    one genome end to end through the drop-in boundary: ingest of Lead objects, upload, pass, the SVCall objects
  cpu_baseline  kind "reference": the UNMODIFIED reference (oracle/_ref, byte-compiled by oracle/make_ref.py) on the same 24 signature
                tables, one process per contig on this box's cores, with vs_baseline; the C oracle beside it (cpu_baseline.port)
  verified / verified_vs_reference   the HIP results of the exact bench workload compared record by record with the oracle run and
                with what the reference's CallTask.execute keeps

Usage: python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
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


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# rocprofv3 summaries of this round, collected with tools/profile.sh.  They are only quoted when they were measured on the
# kernels this run executes: profiles/r04_profile_meta.json records the hash of sniffles_amd/csrc they belong to.
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
ROCPROF_STATS = os.path.join(ROOT, "profiles", "r04_kernel_stats_default.csv")
PROFILE_META = os.path.join(ROOT, "profiles", "r04_profile_meta.json")
ROCPROF_NAMES = {"e45w_consensus_large": "e45w_consensus<2,", "e45w_consensus_small": "e45w_consensus<1,", "d2w_call": "d2w_call<", "e1w_finalize": "e1w_finalize<",
                 "d1w_refine": "d1w_refine", "d4_coverage": "d4_coverage", "a4_binstats": "a4k_binstats", "a6_scatter": "a6k_scatter", "e4c_copy": "e4c_copy",
                 "a1_keys": "a1_keys", "a0_keep": "a0k_keep", "c1_mergeruns": "c1_mergeruns", "d3_rnames": "d3rk_rnames", "b1_seedmetrics": "b1k_seedmetrics",
                 "d5w_covsum": "d5w_covsum", "f4_emit": "f4w_emit", "f5_alt": "f5w_alt", "f3_rank": "f3k_rank", "w1_hist": "w1_hist", "w3_scatter": "w3_scatter",
                 "w4_local": "w4_local<", "w6_emit": "w6_emit<", "d2g_call8": "d2g_call<8"}


def committed_profiles():
    """(rocprofv3 average ms per kernel name, PMC traffic per kernel, note) of the committed round-4 profiles - or empty dicts and
    the reason when they belong to other kernel sources than the ones built here."""
    try:
        from sniffles_amd import build
        meta = json.load(open(PROFILE_META))
        if meta.get("csrc_sha") != build._lib_digest():
            return {}, {}, "profiles/r04_* were collected on other kernel sources (stale): not quoted"
        import csv
        avg = {}
        for r in csv.DictReader(open(ROCPROF_STATS)):
            for short, pat in ROCPROF_NAMES.items():
                if pat in r["Name"].replace("snf::", "").replace(" ", "").replace("void", "") or pat in r["Name"]:
                    avg.setdefault(short, float(r["AverageNs"]) / 1e6)
        pmc = json.load(open(PMC_FILE))["kernels"] if os.path.exists(PMC_FILE) else {}
        return avg, pmc, "rocprofv3 --kernel-trace --stats of the default command, profiles/r04_kernel_stats_default.csv (same kernel sources: hash checked)"
    except Exception as e:  # noqa: BLE001
        return {}, {}, f"no committed profile for these sources ({type(e).__name__})"


def pcie_d2h_peak_gbs(torch, nbytes=32 << 20, reps=8):
    """Device -> pinned host copy rate of this box (GB/s): the roof of the result path, measured, not assumed."""
    if EMU:
        return float("nan")
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    for _ in range(2):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 30 (config 4: 3)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (config 4: 1)")
    ap.add_argument("--config", type=int, default=1, choices=[0, 1, 2, 3, 4], help="BASELINE.json configs[i]")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--coverage", type=float, default=None, help="override the workload's coverage (debug; invalid as a result)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every contig (debug only; invalid as a result)")
    ap.add_argument("--genomes", type=int, default=1,
                    help="genome replicas per batch and rank (SURVEY.md 8d scale knob; the headline configuration is 1)")
    ap.add_argument("--output", choices=["execute", "candidates"], default="execute",
                    help="what a pass returns to the host: execute = what the reference's CallTask.execute returns (QC-passing calls, "
                         "per task sorted by position, parallel.py:265-271), filtered / sorted / compacted on the device; candidates = "
                         "every candidate record (what Task.finalize_candidates returns; the --snf shape)")
    ap.add_argument("--no-configs", action="store_true", help="skip the compact block of the other BASELINE configs (default run only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the all-cores oracle run (and with it --verify)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-reference-baseline", action="store_true",
                    help="skip timing the unmodified reference (oracle/_ref) on the host cores; cpu_baseline.kind is then \"port\"")
    ap.add_argument("--no-wall-clock", action="store_true")
    ap.add_argument("--samples", type=int, default=10, help="config 4: samples of the population")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("SNF_BENCH_INFLIGHT", "2")),
                    help="batches in flight per GPU (host threads, each with its own batch handle and streams): the "
                         "device->host copies, host waits and launch-bound phases of one pass overlap the kernels of "
                         "the other.  1 = strictly one pass at a time.  (Measured in round 3, same box, three alternations: 2 in "
                         "flight 1.87-1.88 ms per step, 3 in flight 1.94-1.99 - a third pass only adds contention)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the result): libraries that print banners through C stdio (RCCL prints its version
    # block to stdout, flushed only at exit when stdout is a pipe) are sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    emu_lib()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not EMU and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    set_dev(torch, local_rank)
    numa = bind_to_gpu_numa(torch, local_rank) if not EMU else "emulation"
    # SNF_BENCH_FORCE_DIST=1: run the whole collective path (RCCL process group, gathers from the worker threads) with a
    # single rank - a dry run of the N > 1 code on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SNF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if EMU:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    ctx = dict(args=args, rank=rank, world=world, local_rank=local_rank, use_dist=use_dist, numa=numa)
    if args.config != 4:
        args.steps = 30 if args.steps is None else args.steps
        args.warmup = 3 if args.warmup is None else args.warmup
    if args.config == 4:
        from tools import bench_population
        out = bench_population.run(ctx)
    else:
        out = run_calling(ctx)
        # N > 1, default (weak) scaling: the SAME line also carries the strong-scaling value - ONE genome over the N ranks (north_star's
        # "30x whole genome at 1/2/4/8 MI355X") - measured by a second run of the same command's ranks over the contig-set queue
        if world > 1 and args.scaling == "weak" and os.environ.get("SNF_BENCH_NO_STRONG") != "1":
            import copy
            a2 = copy.copy(args)
            a2.scaling, a2.no_cpu_baseline, a2.no_wall_clock, a2.no_configs = "strong", True, True, True
            try:
                out2 = run_calling(dict(ctx, args=a2))
                if rank == 0:
                    out["strong"] = dict(value=out2["value"], unit=out2["unit"], ms_per_step=out2["ms_per_step"], steps=out2["steps"], scaling="strong",
                                         signatures=out2["config"]["signatures"], calls=out2["config"]["calls"], parallelism=out2["config"]["parallelism"],
                                         gathered_on_rank0=out2["config"]["gathered_on_rank0"], ranks_seen=out2.get("ranks_seen"),
                                         sets_served=out2.get("sets_served"),
                                         note="ONE genome over the ranks of this run (one LPT-balanced contig set per rank and pass, claimed from the "
                                              "work queue); `value` of the line is the weak-scaling figure (N genome replicas)")
            except Exception as e:  # noqa: BLE001 - the weak line stands on its own
                if rank == 0:
                    out["strong"] = dict(error=f"{type(e).__name__}: {str(e)[:400]}")
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


# ======================================================================================================== configs 0-3
def run_calling(ctx):
    import threading

    import torch
    import torch.distributed as dist

    from sniffles_amd import abi, dist as sdist, lib, synth
    from sniffles_amd.config import SnifflesConfig

    args, rank, world, local_rank, use_dist = (ctx[k] for k in ("args", "rank", "world", "local_rank", "use_dist"))
    wl = WORKLOADS[args.config]
    cfg = SnifflesConfig(**wl["cfg"])
    if os.environ.get("SNF_BENCH_CFG"):   # ablation experiments only (e.g. '{"symbolic": true}'): the line says so and is no result
        for k, val in json.loads(os.environ["SNF_BENCH_CFG"]).items():
            setattr(cfg, k, val)
    strong = args.scaling == "strong"
    G = max(1, args.genomes)
    t0 = time.time()
    if strong:
        # ONE genome; the contigs are partitioned longest-first into one contig set per GPU (LPT-balanced: a claim is ONE device
        # batch of several contigs, which amortises the launch chain of a pass); every rank holds every set resident in HBM so
        # that it can serve whichever it claims from the queue
        specs = task_specs(args, wl, 0, 0, 1)
        # one contig SET per rank and pass: at one rank a set is the whole genome and the W host threads pipeline the passes exactly
        # as the N = 1 line does (measured: two half-genome sets per pass cost 2.44 ms per genome against 1.62 - a pass of half the
        # size is only 1.6x shorter, the launch chain does not shrink); SNF_BENCH_SETS_PER_RANK overrides
        n_groups = max(1, min(len(specs), int(os.environ.get("SNF_BENCH_SETS_PER_RANK", "1")) * world))
        weights = [kw["contig_len"] for _, kw in specs]
        groups = [g for g in sdist.shard_lpt(weights, n_groups) if g]
        group_tasks = [[synth.gen_task(**specs[i][1]) for i in sorted(g)] for g in groups]
        tasks = [t for gt in group_tasks for t in gt]
    else:
        # weak: N genome replicas, (replica, contig) tasks longest-processing-time-first over the ranks
        items = [(rep, ci, kw) for rep in range(world) for ci, kw in task_specs(args, wl, rep, 0, world)]
        mine = sdist.shard_lpt([kw["contig_len"] for _, _, kw in items], world)[rank]
        tasks, task_keys = [], []
        for g in range(G):
            for i in mine:
                rep, ci, _ = items[i]
                kw = dict(task_specs(args, wl, rep, g, world)[ci][1])
                tasks.append(synth.gen_task(**kw))
                task_keys.append(ci)      # batch order is longest-contig-first, not contig order
        group_tasks = [tasks]
    n_sig = sum(t.n_leads for t in tasks)
    n_reads = sum(t.n_reads for t in tasks)
    seq_bytes = sum(int(t.seq_pool.nbytes) for t in tasks)
    t_gen = time.time() - t0

    W = max(1, args.inflight)
    t0 = time.time()
    # handles[w][g]: batch handle of group g for host thread w (same input, independent handles)
    handles = [[lib.Batch(cfg, gt, device=(0 if EMU else local_rank)) for gt in group_tasks] for _ in range(W)]
    t_upload = time.time() - t0
    # N > 1 on one node: every rank's kernels store the result into a shared-memory segment the parent rank maps as well
    # (sniffles_amd.dist.SharedLanding: N PCIe links in parallel, the gather exchanges layouts only).  SNF_BENCH_GATHER=rccl (and
    # --scaling strong) take the block gather over RCCL instead: result blocks in HBM, dist.gather onto rank 0, landed there.
    shared = use_dist and not strong and os.environ.get("SNF_BENCH_GATHER", "shared") != "rccl"
    strong_shared = use_dist and strong and os.environ.get("SNF_BENCH_GATHER", "shared") != "rccl"
    out_mode = (abi.OUT_EXECUTE if args.output == "execute" else abi.OUT_CANDIDATES) | (abi.OUT_DEVICE if use_dist and not shared and not strong_shared else 0)
    for hs in handles:
        for bb in hs:
            bb.set_output(out_mode)
    batches = [h[0] for h in handles]
    if os.environ.get("SNF_BENCH_RESMEM") and not use_dist:     # measurement: results into caller memory (numpy heap / a /dev/shm file)
        import numpy as _np
        _keep = []
        for w_, h_ in enumerate(handles):
            if os.environ["SNF_BENCH_RESMEM"] == "shm":
                path = "/dev/shm/snf_bench_resmem_%d_%d" % (os.getpid(), w_)
                with open(path, "wb") as f_:
                    f_.truncate(96 << 20)
                m_ = _np.memmap(path, _np.uint8, "r+", shape=(96 << 20,)); os.unlink(path)
            else:
                m_ = _np.zeros(96 << 20, _np.uint8)
            _keep.append(m_)
            h_[0].set_result_memory(m_[:48 << 20], m_[48 << 20:])
    handles_box = [handles]

    # capacity of a send buffer (bytes of one result block [records | read names | ALT bytes]): from the first pass, the same
    # on every rank (the largest), with headroom
    n_send = 2 if strong else (2 * W if shared else W)
    task_ids_local = [t.task_id for t in tasks]
    landing = None
    NGEN = 3      # strong: segment generations (pass p writes generation p % 3 once its own gather of pass p - 2 has completed)
    if shared or strong_shared:
        need_b = need_a = 0
        for probe in (handles[0] if strong_shared else handles[0][:1]):     # strong: the largest contig set sizes the segments
            probe.call_candidates(); probe.finalize()
            res0 = probe.fetch(1)
            need_b = max(need_b, 256 + len(res0.calls) * abi.CALL_DTYPE.itemsize + 4 * len(res0.rnames)); need_a = max(need_a, len(res0.alt_pool))
        need = torch.tensor([need_b, need_a], dtype=torch.int64, device=DEV)
        dist.all_reduce(need, op=dist.ReduceOp.MAX)
        # /dev/shm must hold every rank's segments (containers often cap it): otherwise the block gather over RCCL is taken
        n_slots = NGEN * W * len(group_tasks) if strong_shared else 2 * W
        seg_bytes = n_slots * (int(need[0]) * 3 // 2 + int(need[1]) * 3 // 2 + (2 << 20) + 8192)
        try:
            st = os.statvfs("/dev/shm")
            room = st.f_bavail * st.f_frsize >= int(world * seg_bytes * 1.25) and os.environ.get("SNF_BENCH_SHM_FULL") != "1"   # (SNF_BENCH_SHM_FULL=1: the test of this fallback)
        except OSError:
            room = False
        ok_t = torch.tensor([1 if room else 0], dtype=torch.int64, device=DEV)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if not int(ok_t.item()):
            if rank == 0:
                print(f"[bench] /dev/shm cannot hold {world} x {seg_bytes >> 20} MiB of result segments: gathering the blocks over RCCL instead", file=sys.stderr)
            shared = strong_shared = False
            out_mode |= abi.OUT_DEVICE
            for hs in handles:
                for bb in hs:
                    bb.set_output(out_mode)
    if shared or strong_shared:
        # two segments per handle, used in turn: pass k + 1 of a handle writes one while the parent still indexes the other
        # the layouts travel over a host-side (gloo) group: a 72-byte collective must not queue behind the passes' kernels
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: loopback (the box's hostname need not resolve)
        meta_group, meta_dev = None, DEV
        try:
            import datetime
            meta_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
            probe_t = torch.zeros(1, dtype=torch.int64)
            dist.all_reduce(probe_t, group=meta_group)
            meta_dev = None
        except Exception as e:                               # no host-side group here: the layouts go over the default group
            print(f"[bench] gloo group for the layouts unavailable ({type(e).__name__}: {e}); using the default group", file=sys.stderr)
            meta_group, meta_dev = None, DEV
        landing = sdist.SharedLanding(slots=n_slots, block_bytes=int(need[0]) * 3 // 2 + (1 << 20), alt_bytes=int(need[1]) * 3 // 2 + (1 << 20), group=meta_group)
        if strong_shared:
            # segment of (generation, host thread, contig set): every handle stores into memory of its own
            ngs = len(group_tasks)
            slot_of = lambda gen, w, g: (gen * W + w) * ngs + g     # noqa: E731
            for w in range(W):
                for g in range(ngs):
                    for gen in reversed(range(NGEN)):
                        handles[w][g].set_result_memory(*landing.memory(slot_of(gen, w, g)))     # (page-locked here, once)
            set_ids = [[t.task_id for t in gt] for gt in group_tasks]
        else:
            for w in range(W):
                for k in (1, 0):
                    handles[w][0].set_result_memory(*landing.memory(2 * w + k))     # (page-locked here, once)
            ids_all = [None] * world
            dist.all_gather_object(ids_all, task_ids_local)
    if use_dist and not shared and not strong_shared:
        probe = handles[0][0]
        probe.call_candidates(); probe.finalize()
        res0 = probe.fetch(1)
        blk = 256 * 3 + len(res0.calls) * abi.CALL_DTYPE.itemsize + 4 * len(res0.rnames) + len(res0.alt_pool)
        cap_t = torch.tensor([blk * (len(group_tasks) if strong else 1) * 3 // 2 + (1 << 20)], dtype=torch.int64, device=DEV)
        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        cap_bytes = int(cap_t.item())
        sends = [torch.zeros(cap_bytes, dtype=torch.uint8, device=DEV) for _ in range(n_send)]
        recv = torch.empty(world * cap_bytes, dtype=torch.uint8, device=DEV) if rank == 0 else None
        ids_all = [None] * world
        dist.all_gather_object(ids_all, task_ids_local)          # once: which rank holds which tasks (batch order)
    # Collectives run on ONE communication thread per rank, in the order the passes finish: a worker thread exports its
    # result block (device-to-device) into a send buffer, queues it and goes on with its next pass; the gather
    # (sniffles_amd.dist.gather_results: layouts, then the blocks onto rank 0, merged there by task id) overlaps that pass.
    # Every rank issues the same sequence of collectives, so the order matches everywhere.
    import queue
    comm_q = queue.Queue()
    send_free = [threading.Event() for _ in range(n_send)]
    for ev in send_free:
        ev.set()
    comm_err = []
    gathered_box = [None]

    def comm_loop():
        try:
            set_dev(torch, local_rank)
            while True:
                item = comm_q.get()
                if item is None:
                    comm_q.task_done()
                    return
                s, lay, ids = item
                if strong_shared:      # s = (entries of this rank in the pass, event to set)
                    g = sdist.gather_sets_shared(landing, s[0], len(group_tasks), set_ids, group=meta_group, device=meta_dev)
                    if g is not None:
                        gathered_box[0] = g
                    s[1].set()
                    comm_q.task_done()
                    continue
                if shared:
                    g = sdist.gather_results_shared(landing, s, lay, ids, group=meta_group, task_ids_per_rank=ids_all, device=meta_dev)
                else:
                    g = sdist.gather_results(sends[s], lay, ids, dst=0, recv_buffer=recv,
                                             task_ids_per_rank=None if strong else ids_all)
                if g is not None:
                    gathered_box[0] = g
                send_free[s].set()
                comm_q.task_done()
        except BaseException as e:  # noqa: BLE001 - re-raised in the main thread
            comm_err.append(e)
            for ev in send_free:
                ev.set()
            while True:   # keep draining so that queue.join() cannot hang
                try:
                    comm_q.get_nowait(); comm_q.task_done()
                except queue.Empty:
                    break

    comm_thread = None
    if use_dist:
        comm_thread = threading.Thread(target=comm_loop, daemon=True)
        comm_thread.start()

    phase_s = [0.0, 0.0, 0.0, 0.0]
    lay_box = [None] * W
    turn_box = [0] * W

    def one_pass(w, g=0):
        """One full pass of the hot path over group g on thread w's handle: candidates, finalize, D2H of the results."""
        batch = handles_box[0][w][g]
        t_a = time.perf_counter()
        batch.run_pass()                      # call_candidates + finalize as one unit (snf_batch_pass: a HIP graph from the second pass on)
        t_b = time.perf_counter()
        t_c = t_b
        # the one host wait of the pass: the result block [records | read names | ALT bytes] is in pinned host memory when this
        # returns (N > 1: it stays in HBM for the gather, the export below is the wait)
        n = batch.fetch_raw(1) if not use_dist else 0
        if shared:
            lay_box[w] = batch.fetch_layout()      # (the same one host wait: the result lies in this handle's shared segment)
        t_d = time.perf_counter()
        if w == 0:
            phase_s[0] += t_b - t_a; phase_s[1] += t_c - t_b; phase_s[2] += t_d - t_c; phase_s[3] += 1

        return n

    def barrier():
        if use_dist:
            comm_q.join()                                              # every queued gather has completed
            if comm_err:
                raise comm_err[0]
            dist.barrier()
        dev_sync(torch)

    n_calls_box = [0]
    served_box = [0]          # device batches (weak: passes, strong: contig sets) this rank ran inside the timed region

    def run_threads(fn_of_w):
        if W == 1:
            fn_of_w(0)
            return
        errs = []

        def worker(w):
            try:
                set_dev(torch, local_rank)
                fn_of_w(w)
            except BaseException as e:  # noqa: BLE001 - re-raised in the main thread
                errs.append(e)
                pass_barrier.abort()

        ths = [threading.Thread(target=worker, args=(w,)) for w in range(W)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    def run_passes_weak(total):
        """Exactly `total` passes, split evenly over the W host threads (thread w works on its own batch handle)."""
        def body(w):
            for _ in range(total // W + (1 if w < total % W else 0)):
                if shared:
                    slot = 2 * w + (turn_box[w] & 1); turn_box[w] += 1
                    send_free[slot].wait()                             # the parent has indexed the result this segment held two passes ago
                    send_free[slot].clear()
                    handles_box[0][w][0].set_result_memory(*landing.memory(slot))
                    one_pass(w)
                    served_box[0] += 1
                    n_calls_box[0] = lay_box[w]["n_calls"]
                    comm_q.put((slot, lay_box[w], task_ids_local))
                    continue
                n_calls_box[0] = one_pass(w)
                served_box[0] += 1
                if use_dist:
                    send_free[w].wait()                                # the previous gather of this handle has left the buffer
                    send_free[w].clear()
                    lay = batches[w].export_device(sends[w].data_ptr(), cap_bytes)     # (blocks until the block is there)
                    n_calls_box[0] = lay["n_calls"]
                    comm_q.put((w, lay, task_ids_local))
        run_threads(body)

    pass_barrier = threading.Barrier(W)
    ng = len(group_tasks)
    gw = [sum(t.n_leads for t in gt) for gt in group_tasks]

    def run_passes_strong(total):
        """`total` passes over the ONE genome, one after the other.  Per pass a work queue of the contig groups, heaviest
        first (sniffles_amd.dist.TaskQueue: one atomic add on the process group's store per claim; a plain counter for a
        single rank); the W host threads of every rank claim groups and run them on their own handle of the group.  The
        call records of the groups a rank served are packed into one send buffer and gathered on rank 0 once per pass,
        on the communication thread, while the next pass is already running."""
        lock = threading.Lock()
        if use_dist:
            queues = [sdist.TaskQueue(gw, key="snf_bench", barrier=False) for _ in range(total)]   # same keys on every rank
            dist.barrier()
        else:
            queues = [sdist.LocalQueue(gw) for _ in range(total)]
        acc = [[]]
        ident = list(range(max(t.task_id for t in tasks) + 1))

        def body(w):
            for p in range(total):
                if use_dist and w == 0:
                    send_free[p % 2].wait()                            # the gather of pass p - 2 has left the buffer
                    send_free[p % 2].clear()
                    acc[0] = []
                if W > 1:
                    pass_barrier.wait()
                for g in queues[p]:
                    one_pass(w, g)
                    served_box[0] += 1
                    if use_dist:                                       # the blocks of the groups this rank served are merged on its host
                        blk = sdist.result_block(handles[w][g].fetch(1))
                        with lock:
                            acc[0].append((blk, [t.task_id for t in group_tasks[g]]))
                if W > 1:
                    pass_barrier.wait()                                # every thread of this rank is through pass p
                if use_dist and w == 0:
                    local = sdist.merge_blocks([b_ for b_, _ in acc[0]], [i_ for _, i_ in acc[0]])
                    lay, blob = sdist.result_block(local)
                    if lay["bytes"] > cap_bytes:
                        raise RuntimeError("send buffer too small")
                    sends[p % 2][:lay["bytes"]].copy_(torch.from_numpy(blob))
                    comm_q.put((p % 2, lay, ident))
        run_threads(body)

    def run_passes_strong_shared(total):
        """`total` passes over the ONE genome, pipelined: per pass a work queue of the contig SETS, heaviest first
        (`dist.TaskQueue`: one atomic add on the process group's store per claim - issued while the previous claim's kernels run);
        a claimed set is ONE device batch whose result the kernels store into this rank's shared-memory segment of (generation,
        thread, set) - exactly the weak-scaling landing.  When the threads of a rank are through a pass, its communication
        thread issues the pass's one collective (`dist.gather_sets_shared`: layouts only) while the threads are already in the
        next pass; rank 0 then holds the whole genome's result, tasks in id order, in place.  No host copy, no merge."""
        queues = [sdist.TaskQueue(gw, key="snf_bench", barrier=False) for _ in range(total)]   # same keys on every rank
        dist.barrier()
        lock = threading.Lock()
        entries = [[] for _ in range(total)]
        through = [0] * total
        done = [threading.Event() for _ in range(total)]

        def body(w):
            nxt = queues[0].claim()
            for p in range(total):
                if p >= NGEN - 1:
                    done[p - (NGEN - 1)].wait()                        # rank 0 has indexed what generation p % NGEN held
                    if comm_err:
                        return
                g = nxt
                while g is not None:
                    bb = handles[w][g]
                    t_a = time.perf_counter()
                    bb.set_result_memory(*landing.memory(slot_of(p % NGEN, w, g)))
                    bb.run_pass()                                      # (enqueues only)
                    t_b = time.perf_counter()
                    nxt = queues[p].claim()                            # the next claim travels while the kernels run
                    t_c = time.perf_counter()
                    lay = bb.fetch_layout()                            # the one host wait: the result lies in the segment
                    t_d = time.perf_counter()
                    if w == 0:
                        phase_s[0] += t_b - t_a; phase_s[1] += t_c - t_b; phase_s[2] += t_d - t_c; phase_s[3] += 1
                    with lock:
                        entries[p].append((slot_of(p % NGEN, w, g), lay, g))
                        served_box[0] += 1
                    g = nxt
                nxt = queues[p + 1].claim() if p + 1 < total else None
                with lock:
                    through[p] += 1
                    last = through[p] == W
                if last:
                    comm_q.put(((entries[p], done[p]), None, None))
        run_threads(body)

    run_passes = (run_passes_strong_shared if strong_shared else run_passes_strong) if strong else run_passes_weak
    run_passes(max(1, args.warmup) if strong else -(-args.warmup // W) * W)  # >= warmup passes, the same number on every handle
    barrier()
    import gc
    gc.collect(); gc.disable()           # no collector pauses inside the ~50 ms that are timed
    batches[0].timings_mean_reset()       # the library averages every kernel's HIP-event duration over the passes from here on
    served_box[0] = 0
    t0 = time.perf_counter()
    run_passes(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    served = served_box[0]
    ranks_seen, sets_served = 1, [served]
    if use_dist:              # who took part, and how the device batches of the timed region were spread over the ranks
        ranks_seen = dist.get_world_size()
        sets_served = [None] * ranks_seen
        dist.all_gather_object(sets_served, served)
    if shared and gathered_box[0] is not None:
        gathered_box[0].detach()          # (the passes below write the segments again)
    n_calls = n_calls_box[0]
    # per-kernel HIP-event times (on the kernels' own streams), AVERAGED over handle 0's passes inside the timed region
    timings = batches[0].timings_mean() or batches[0].timings()
    # reference point outside the timed region: the same pass with ONE batch in flight (per-pass latency)
    lat_ms = None
    if W > 1 and not strong:
        barrier()
        batches[0].timing_every(1)        # (every one of these passes carries the event brackets: their times are reported as such)
        t1 = time.perf_counter()
        for _ in range(5):
            one_pass(0)
        dev_sync(torch)
        lat_ms = (time.perf_counter() - t1) / 5 * 1e3
    barrier()   # also drains the communication thread before the main thread issues collectives again
    timings_alone = dict((k[0], k[1]) for k in batches[0].timings()) if lat_ms else {}
    # for the record: the same passes when every call_candidates ALSO rebuilds the read index (sorted read ends, hap prefix
    # counts - the device form of the coverage vector, which the reference builds during extraction and the library at upload)
    ms_with_index = None
    if world == 1 and not strong and not use_dist and not args.no_wall_clock:
        os.environ["SNF_READPREP_EACH_PASS"] = "1"
        extra = [[lib.Batch(cfg, tasks, device=(0 if EMU else local_rank))] for _ in range(W)]
        del os.environ["SNF_READPREP_EACH_PASS"]
        handles_box[0] = extra
        k2 = max(W, args.steps // 2)
        run_passes_weak(W); dev_sync(torch)
        t1 = time.perf_counter(); run_passes_weak(k2); dev_sync(torch)
        ms_with_index = (time.perf_counter() - t1) / k2 * 1e3
        handles_box[0] = handles
        for hs in extra:
            hs[0].close()

    tt = torch.tensor([dt], dtype=torch.float64, device=DEV)
    tot = torch.tensor([n_sig, n_calls], dtype=torch.int64, device=DEV)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if not strong:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    total_sig, total_calls = int(tot[0].item()), int(tot[1].item())
    if strong:
        total_sig = n_sig                    # every rank holds the same ONE genome
        res_calls = 0
        for g in range(len(group_tasks)):   # calls of the whole genome (every group once, outside the timed region)
            n_g = one_pass(0, g)
            res_calls += int(len(handles[0][g].fetch(1).calls)) if use_dist else n_g
        barrier()
        total_calls = res_calls

    out = None
    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = total_sig * args.steps / dt_max
        # dominant kernel: the largest average launch duration over the timed passes of handle 0 (HIP events around each launch)
        kern = sorted(timings, key=lambda x: -x[1])
        # dominant KERNEL: entries that bracket a sequence of library launches (rocPRIM sort / scan passes) or a copy are
        # listed in top_kernels but are not a kernel whose roofline could be stated
        single = [k for k in kern if not k[0].startswith(("sort_", "scan_", "d2h_", "front_"))]      # (front_window brackets the six launches of the window front end)
        top = single[0] if single else ("none", 0.0, 0)
        gpu_ms = sum(k[1] for k in kern)
        achieved = (top[2] / (top[1] * 1e-3)) / 1e9 if top[1] > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC counters can not be read from inside the process; they were collected
        # with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes) on configs[1] and are kept, corrected as
        # the microarch guide prescribes, under profiles/
        traffic = None
        prof_avg, pmc, prof_note = committed_profiles()
        same_workload = args.scale == 1.0 and args.coverage is None and args.config == 1 and not strong and world == 1 and W == 2
        if same_workload and top[0] in pmc:
            traffic = pmc[top[0]]["hbm_bytes"]
        rocprof_ms = prof_avg.get(top[0]) if same_workload else None
        # result path: what one pass sends to the host over PCIe, against the measured device -> pinned-host copy rate of this box
        res_one = batches[0].fetch(1) if not use_dist else None
        result_bytes = (len(res_one.calls) * abi.CALL_DTYPE.itemsize + 4 * len(res_one.rnames) + len(res_one.alt_pool)) if res_one is not None else None
        pcie_peak = pcie_d2h_peak_gbs(torch)
        result_path = dict(bound="pcie", what="bytes of the result block one pass hands to the host (records + read names + ALT bytes of the "
                           + ("calls CallTask.execute keeps" if args.output == "execute" else "candidates") + "), stored by the kernels straight into pinned "
                           "host memory, over the time of a step; peak = device -> pinned host copy rate measured in this run",
                           bytes_per_pass=result_bytes, records_per_pass=(len(res_one.calls) if res_one is not None else None),
                           achieved=(round(result_bytes / (ms_per_step * 1e-3) / 1e9, 2) if result_bytes else None), peak=round(pcie_peak, 2), unit="GB/s",
                           frac=(round(result_bytes / (ms_per_step * 1e-3) / 1e9 / pcie_peak, 4) if result_bytes else None))
        # whole pass against the roofline: SURVEY.md 8(d) algorithmic bytes of one pass (72 B/signature + consensus bytes as
        # counted by the kernels + 8 B/read + 20 B/call) over the time of one pass
        cons_bytes = sum(k[2] for k in kern if k[0].startswith(("e45w_consensus", "e4c_copy")))
        pass_bytes = 72 * n_sig + cons_bytes + 8 * n_reads + 20 * n_calls
        # The stage is latency-bound (dependent LDS / L2 round trips at 2-5 waves per SIMD), far below either roof: the HBM
        # fraction is stated because the data is HBM-resident integer / byte work (no MFMA), the PCIe fraction of the result
        # path sits next to it in `result_path`
        roofline = dict(bound="hbm", kernel=top[0], achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                        kernel_ms=round(top[1], 4), algorithmic_bytes=int(top[2]),
                        kernel_ms_source="mean HIP-event duration of the kernel's launches (events on the stream it is launched on) over the timed passes of handle 0 (batches in flight as configured); "
                                         "the LARGE consensus kernel is bracketed on every pass, the other kernels on every 8th pass of the handle (snf_batch_timing_every)",
                        rocprof_ms=(round(rocprof_ms, 4) if rocprof_ms else None),
                        rocprof_frac=(round(top[2] / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if rocprof_ms else None),
                        profile_note=prof_note, result_path=result_path,
                        gpu_ms_all_kernels=round(gpu_ms, 3),
                        whole_pass=dict(algorithmic_bytes=int(pass_bytes),
                                        achieved=round(pass_bytes / (ms_per_step * 1e-3) / 1e9, 2) if not strong and world == 1 else None,
                                        frac=round(pass_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if not strong and world == 1 else None),
                        top_kernels=[dict(name=k[0], ms=round(k[1], 4), algorithmic_bytes=int(k[2]),
                                          **({"rocprof_ms": round(prof_avg[k[0]], 4)} if same_workload and k[0] in prof_avg else {}),
                                          **({"ms_one_batch_in_flight": round(timings_alone[k[0]], 4)} if k[0] in timings_alone else {}))
                                     for k in kern[:int(os.environ.get("SNF_BENCH_TOPK", "8"))]])
        n_contigs = len(wl["contigs"] or synth.CONTIGS)
        out = dict(metric="SV-signatures clustered/sec (clustering + calling + QC + genotype + INS consensus)",
                   value=value, unit="signatures/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_per_step, higher_is_better=True, scaling=args.scaling, vs_baseline=None, dtype="int32/f64",
                   ranks_seen=ranks_seen, sets_served=sets_served,
                   data="synthetic" + (" - SNF_BENCH_EMU=1: host emulation of the kernels over gloo, a test of the N > 1 plumbing, NOT a result" if EMU else "")
                   + (" - SNF_BENCH_CFG overrides the workload's configuration: an ablation, NOT a result" if os.environ.get("SNF_BENCH_CFG") else ""),
                   config=dict(workload=wl["name"] + ", synthetic signature tables (SURVEY.md 8d)", baseline_config=args.config,
                               replicas=1 if strong else world, genomes_per_batch=G,
                               tasks=n_contigs * (1 if strong else world) * G,
                               coverage=args.coverage if args.coverage is not None else wl["coverage"], scale=args.scale,
                               signatures=total_sig, reads_rank0=n_reads, ins_seq_bytes_rank0=seq_bytes,
                               calls=total_calls,
                               parallelism=(f"one genome, {len(group_tasks)} contig sets (LPT-balanced, one device batch each) claimed from a shared work queue by {world} ranks x {W} host threads, passes pipelined"
                                            if strong else f"contig-sharded x{world}") + (", every rank's result stored into node-shared host memory by its own kernels, layouts gathered on rank 0 (dist.SharedLanding)" if (shared or strong_shared) else ", RCCL gather of the result blocks on rank 0" if use_dist else ", one process: no gather"),
                               batches_in_flight_per_gpu=W, host_binding=ctx.get("numa"),
                               gathered_on_rank0=(dict(ranks=world, records=int(len(gathered_box[0].calls)), alt_bytes=int(len(gathered_box[0].alt_pool)),
                                                       read_names=int(len(gathered_box[0].rnames)), tasks=int(len(gathered_box[0].task_ids)),
                                                       order="task id, then position (parallel.py:270-271, sniffles:544)")
                                                  if use_dist and gathered_box[0] is not None else None),
                               ms_per_pass_one_batch_in_flight=(round(lat_ms, 3) if lat_ms else None),
                               output=args.output + (": what CallTask.execute returns (parallel.py:265-271) - QC-passing calls, per task sorted by position, "
                                                     "filtered / sorted / compacted on the device" if args.output == "execute" else ": every candidate record"),
                               timed_region="call_candidates + finalize (both enqueue only) + the one host wait of the pass, after which the result block "
                                            "(records, read names, ALT bytes) is in pinned host memory; the read index (sorted read "
                                            "ends + hap prefix counts = the coverage vector / hap tables the reference builds during "
                                            "extraction, excluded from cpu_baseline as well) is built once at upload",
                               ms_per_step_with_read_index_rebuilt_every_pass=(round(ms_with_index, 3) if ms_with_index else None),
                               gen_s=round(t_gen, 2), upload_s=round(t_upload, 2),
                               host_ms_per_step=dict(enqueue_call_candidates=round(phase_s[0] / max(1, phase_s[3]) * 1e3, 3),
                                                     finalize=round(phase_s[1] / max(1, phase_s[3]) * 1e3, 3),
                                                     fetch_d2h=round(phase_s[2] / max(1, phase_s[3]) * 1e3, 3)),
                               parity_unpinned=["edit distance vs edlib itself (edlib absent; pinned to the exact Levenshtein DP)",
                                                "pysam stand-in of the extraction oracle (pinned by the reference's 17 known-answer reads)"]),
                   roofline=roofline)
        if world == 1 and not strong and G == 1:
            if not args.no_wall_clock:
                out["wall_clock"] = wall_clock(cfg, tasks, local_rank, task_specs(args, wl, 0, 0, 1), wl["cfg"])
            if not args.no_cpu_baseline:
                got = exe = None
                if not args.no_verify:       # every candidate record against the oracle, and the execute-mode block against its definition
                    bb = batches[0]
                    bb.set_output(abi.OUT_CANDIDATES); bb.call_candidates(); bb.finalize(); got = bb.fetch(1)
                    bb.set_output(abi.OUT_EXECUTE); bb.call_candidates(); bb.finalize(); exe = bb.fetch(1)
                base, ver = cpu_baseline_and_verify(args, wl, got, task_keys, exe, cfg)
                out["cpu_baseline"] = base
                if not args.no_reference_baseline:
                    # the UNMODIFIED reference on this box's host cores (oracle/_ref, staged by oracle/make_ref.py), same tables
                    try:
                        ref_base = reference_baseline(args, wl, exe, tasks, task_keys, out)
                    except Exception as e:  # noqa: BLE001 - a baseline that cannot run must not take the line down; it says why
                        ref_base = None
                        base["reference_error"] = f"{type(e).__name__}: {str(e)[:600]}"
                    if ref_base is not None:
                        ref_base["port"] = base          # the C restatement stays beside it
                        out["cpu_baseline"] = ref_base
                    else:
                        base["reference_note"] = ("the staged reference build (oracle/_ref, made by oracle/make_ref.py during build() where "
                                                  "/root/reference exists) is not on this box: kind stays \"port\"")
                if ver is not None:
                    out["verified"] = ver["ok"]
                    out["verify"] = ver
            if args.config == 1 and not args.no_configs and args.scale == 1.0 and args.coverage is None:
                for hs in handles:           # (the headline's batches are done: with their streams alive the small config ran 0.53
                    for bb in hs:            #  instead of 0.35 ms per step - more streams than hardware queues)
                        bb.close()
                out["configs"] = other_configs(ctx)
    if comm_thread is not None:
        comm_q.put(None)
        comm_thread.join(timeout=30)
    for hs in handles:
        for bb in hs:
            bb.close()
    return out


def bind_to_gpu_numa(torch, local_rank):
    """One process per GPU, bound to the CPUs of the NUMA node the GPU hangs off (what a launcher does for every rank): the
    host threads that drive the batches, their pinned buffers and the staging arena then sit next to the PCIe root of the
    device.  (Measured on the 2-socket GPU box: no difference for one rank - four runs each 2.10-2.41 ms unbound, 2.12-2.30
    bound; the run-to-run spread of ~10 % has another cause.  Kept for the N-rank launches.)  Best effort (sysfs); SNF_BENCH_NO_NUMA=1 turns it off.  Returns what was done, for the output line."""
    if os.environ.get("SNF_BENCH_NO_NUMA") == "1":
        return "off"
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return "gpu has no numa node"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"node {node}: no allowed cpu"
        os.sched_setaffinity(0, cpus)
        return f"node {node} ({len(cpus)} cpus)"
    except Exception as e:  # noqa: BLE001 - measurement hygiene only
        return f"unavailable ({type(e).__name__})"


def worker_processes(specs, cfg_kw, device):
    """The per-task seam in the reference's deployment shape (tools/bench_workers.py): P worker processes share this GPU, each runs
    Task.call_candidates + finalize_candidates over its contigs (longest first), two tasks in flight per worker; `leads`: the lead
    providers hold Lead objects (the one walk that turns them into columns is inside call_candidates), `columns`: typed columns."""
    from tools import bench_workers
    out = {}
    plan = [(4, "columns", "api"), (4, "leads", "api"), (8, "columns", "api"), (8, "columns", "execute"), (8, "leads", "api"),
            (24, "columns", "api"), (24, "leads", "api")]
    if os.environ.get("SNF_BENCH_WORKERS"):      # e.g. "8" or "4,24"
        want = {int(x) for x in os.environ["SNF_BENCH_WORKERS"].split(",") if x}
        plan = [p for p in plan if p[0] in want]
    for procs, form, shape in plan:
        key = f"P{procs}_{form}_{shape}"
        try:
            out[key] = bench_workers.run(specs, cfg_kw, procs, form, shape, device)
        except Exception as e:  # noqa: BLE001 - an extra measurement must not take the line down
            out[key] = f"failed: {type(e).__name__}: {str(e)[:300]}"
    out["note"] = ("hot_all_ms = the slowest worker's time over its tasks from a common barrier (inputs built and device context warm before it, as "
                   "oracle/ref_pool.py times the reference); ingest_all_ms = that worker's Lead objects -> columns walk alone; one MI355X shared by all workers")
    return out


def wall_clock(cfg, tasks, device, specs=None, cfg_kw=None):
    """One genome end to end through the drop-in boundary, outside the timed region (milliseconds): `batched` = all
    contig tasks in one device batch (the library's native shape); `per_task_api` = the reference's own call sequence,
    Task.call_candidates + Task.finalize_candidates task by task, SVCall objects out (sniffles_amd.parallel)."""
    from sniffles_amd import lib, parallel, pipeline, sv
    os.environ["SNF_PROF"] = "1"          # the library prints its own split of the upload to stderr
    t0 = time.perf_counter()
    from sniffles_amd import abi
    b = lib.Batch(cfg, tasks, device=device)
    t1 = time.perf_counter()
    del os.environ["SNF_PROF"]
    # what the reference's workers hand to the parent - CallTask.execute's result (parallel.py:264-271): the QC-passing calls of
    # every task sorted by position, filtered and ordered on the device; those (26.8 k of the 94 k candidates) become objects
    b.set_output(abi.OUT_EXECUTE)
    b.call_candidates(); b.finalize(); b.sync()
    t2 = time.perf_counter()
    res = b.fetch(1, copy=False)          # views of the library's pinned result block, as sniffles_amd.parallel.Task reads them
    t3 = time.perf_counter()
    n = 0
    for t, ti in enumerate(tasks):
        lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
        calls = sv.materialize_candidates(res, ti, lo, hi)
        sv.apply_final(calls, res, ti, lo)
        for c in calls:
            c.finalize()
        n += len(calls)
    t4 = time.perf_counter()
    # for the record: every candidate as an object (what Task.finalize_candidates returns with keep_qc_fails - the --snf shape)
    b.set_output(abi.OUT_CANDIDATES)
    b.call_candidates(); b.finalize(); b.sync()
    res_all = b.fetch(1, copy=False)
    tc = time.perf_counter()
    n_all = 0
    for t, ti in enumerate(tasks):
        lo, hi = int(res_all.task_call_off[t]), int(res_all.task_call_off[t + 1])
        calls = sv.materialize_candidates(res_all, ti, lo, hi)
        sv.apply_final(calls, res_all, ti, lo)
        n_all += len(calls)
    all_ms = (time.perf_counter() - tc) * 1e3
    del calls
    b.set_output(abi.OUT_EXECUTE)
    b.call_candidates(); b.finalize(); b.sync()
    res = b.fetch(1, copy=False)
    # the same records as VCF text without the objects in between (vcf.VCF.write_records, the BAM -> VCF flow with objects=False)
    vcf_ms, vcf_bytes = None, None
    try:
        import io
        import numpy as np
        from sniffles_amd import vcf
        if not getattr(cfg, "sample_ids_vcf", None):
            cfg.sample_ids_vcf = [(0, "SAMPLE")]
        buf = io.StringIO()
        w = vcf.VCF(cfg, buf)
        if w.can_write_records():
            tv = time.perf_counter()
            for t, ti in enumerate(tasks):
                lo, hi = int(res.task_call_off[t]), int(res.task_call_off[t + 1])
                keep = lo + np.flatnonzero(res.calls["qc"][lo:hi] != 0)
                keep = keep[np.argsort(res.calls["pos"][keep], kind="stable")]
                w.write_records(res, ti, keep)
            vcf_ms, vcf_bytes = round((time.perf_counter() - tv) * 1e3, 2), len(buf.getvalue())
    except Exception as e:                      # never let the extra measurement take the bench line down
        vcf_ms = f"failed: {type(e).__name__}: {e}"
    b.close()
    # the same upload again: the batch above has returned its device slab to the library's cache, this one reuses it.  An upload
    # that has to hipMalloc its slab (the first batch of a process, or one created while the earlier batches are alive - the case
    # above, behind the two timed batches) pays 60-200 ms for the allocation; a pipeline pays that once per live batch
    tw = time.perf_counter()
    b2 = lib.Batch(cfg, tasks, device=device)
    warm_ms = (time.perf_counter() - tw) * 1e3
    b2.close()
    batched = dict(vcf_text_from_records_ms=vcf_ms, vcf_text_bytes=vcf_bytes, upload_ms=round((t1 - t0) * 1e3, 2), pass_ms=round((t2 - t1) * 1e3, 2), d2h_ms=round((t3 - t2) * 1e3, 2),
                   materialise_ms=round((t4 - t3) * 1e3, 2), end_to_end_ms=round((t4 - t0) * 1e3, 2), svcalls=n,
                   materialise_all_candidates_ms=round(all_ms, 2), candidates=n_all,
                   upload_GBps=round(_input_bytes(tasks) / max(1e-9, t1 - t0) / 1e9, 2),
                   upload_slab_reused_ms=round(warm_ms, 2), end_to_end_slab_reused_ms=round((t4 - t1) * 1e3 + warm_ms, 2))
    # the reference's worker loop, one process: per contig task the two-call seam (Task.call_candidates + finalize_candidates, every candidate
    # an object) or the one-step drop-in (CallTask.execute_calls: upload, pass, objects of the kept calls).  `pipelined`: the loop keeps two
    # tasks in flight (Task.prepare: task k + 1 uploads and runs on the device while task k's records become objects)
    def per_task(shape, pipelined):
        ts = []
        for ti in tasks:
            task = parallel.CallTask(id=ti.task_id, sv_id=0, contig=ti.contig, start=0, end=ti.contig_len, config=cfg, tandem_repeats=None, device=device)
            task.lead_provider = pipeline._Extracted(ti)
            ts.append(task)
        ex = True if shape == "execute" else None
        ta = time.perf_counter()
        n_ = 0
        if pipelined and ts:
            ts[0].prepare(cfg, execute=ex)
        for k, task in enumerate(ts):
            if pipelined and k + 1 < len(ts):
                ts[k + 1].prepare(cfg, execute=ex)
            if shape == "execute":
                n_ += len(task.execute_calls(cfg))
            else:
                cands = task.call_candidates(False, cfg)
                n_ += len(task.finalize_candidates(cands, True, cfg))
            task.close()
        return (time.perf_counter() - ta) * 1e3, n_
    exe_serial_ms, n3 = per_task("execute", False)
    exe_ms, _ = per_task("execute", True)
    api_serial_ms, n2 = per_task("api", False)
    api_ms, _ = per_task("api", True)
    # the INPUT half of the object boundary: Lead objects -> LeadProvider.record_lead / record_read -> TaskInput columns
    # (leadprov.py:400-418 on the reference's side).  Measured on the smallest contig task of the workload (building the Lead objects
    # themselves is the extraction's work and is not timed); the genome figure is that rate x all signatures
    ingest = None
    try:
        from sniffles_amd import leadprov
        ti_s = min(tasks, key=lambda t: t.n_leads)
        objs = list(leadprov.iter_leads(ti_s))
        rs_, re_, hp_ = ti_s.read_start.tolist(), ti_s.read_end.tolist(), ti_s.read_hp.tolist()
        lp = leadprov.LeadProvider(cfg, 0, ti_s.contig, contig_len=ti_s.contig_len)
        ti0 = time.perf_counter()
        for ld in objs:
            lp.record_lead(ld, 0)
        for a_, b_, c_ in zip(rs_, re_, hp_):
            lp.record_read(a_, b_, c_)
        ti1 = time.perf_counter()
        lp.to_task_input(ti_s.task_id, 0, None, ti_s.qc_nm_threshold)
        ti2 = time.perf_counter()
        n_all_leads = sum(t.n_leads for t in tasks)
        per_lead = (ti2 - ti0) / max(1, ti_s.n_leads)
        ingest = dict(contig=ti_s.contig, leads=int(ti_s.n_leads), reads=int(ti_s.n_reads), record_ms=round((ti1 - ti0) * 1e3, 2),
                      to_task_input_ms=round((ti2 - ti1) * 1e3, 2), us_per_lead=round(per_lead * 1e6, 3),
                      ingest_ms_genome_one_core=round(per_lead * n_all_leads * 1e3, 1),
                      ingest_ms_largest_task=round(per_lead * max(t.n_leads for t in tasks) * 1e3, 1),
                      note="record_lead / record_read append; to_task_input = ONE walk over the Lead objects in C (_snf_fast.lead_columns) + name "
                           "interning; one process per contig in the reference's layout: the largest task bounds the wall clock")
        del objs
    except Exception as e:  # noqa: BLE001
        ingest = f"failed: {type(e).__name__}: {e}"
    workers = None
    if specs is not None and not EMU and os.environ.get("SNF_BENCH_NO_WORKERS") != "1":
        workers = worker_processes(specs, cfg_kw or {}, device)
    return dict(batched=batched, ingest=ingest, worker_processes=workers,
                per_task_api=dict(end_to_end_ms=round(api_ms, 2), one_task_at_a_time_ms=round(api_serial_ms, 2), tasks=len(tasks), svcalls=n2),
                per_task_execute=dict(end_to_end_ms=round(exe_ms, 2), one_task_at_a_time_ms=round(exe_serial_ms, 2), tasks=len(tasks), svcalls=n3),
                note="one genome, inputs in host numpy columns; upload = snf_batch_create + add_task + upload; "
                     "batched = all contig tasks in one device batch, the objects of what CallTask.execute returns (QC-passing calls, sorted; "
                     "materialise_all_candidates_ms: every candidate instead); per_task_api = 24 x Task.call_candidates + finalize_candidates "
                     "(every candidate an object twice over, the reference's two-call shape); per_task_execute = 24 x CallTask.execute_calls; both with two "
                     "tasks in flight (Task.prepare: the next task uploads and runs while this one's records become objects), one_task_at_a_time_ms without; "
                     "d2h = results in the library's pinned block (read in place); materialise = SVCall Python objects (host); "
                     "vcf_text_from_records = the QC-passing records as VCF lines straight from the record table (no objects)")


def _input_bytes(tasks):
    n = 0
    for t in tasks:
        n += sum(int(a.nbytes) for a in t.leads.values()) + int(t.seq_pool.nbytes)
        n += int(t.read_start.nbytes) + int(t.read_end.nbytes) + int(t.read_hp.nbytes)
    return n


def other_configs(ctx) -> dict:
    """The other BASELINE.json configs in the default line (compact: a few steps each, verified against the oracle), so that
    they are measured wherever the headline is: configs[0] (chr20 only), [2] (60x HiFi), [3] (--mosaic) through the same
    passes as the headline, configs[4] (10-sample merge) through tools/bench_population.  Outside the headline's timed region."""
    import copy
    import threading

    import torch

    from sniffles_amd import abi, lib, synth
    from sniffles_amd.config import SnifflesConfig
    args, local_rank = ctx["args"], ctx["local_rank"]
    out = {}
    for k in (0, 2, 3):
        t_all = time.time()
        try:
            wl = WORKLOADS[k]
            a = copy.copy(args); a.config = k
            cfg = SnifflesConfig(**wl["cfg"])
            specs = task_specs(a, wl, 0, 0, 1)
            tasks = [synth.gen_task(**kw) for _, kw in specs]
            W, steps, warm = 2, 12, 2
            if k == 0:
                steps, warm = 48, 8      # (a small batch is replayed as a HIP graph: its first few launches cost milliseconds each - not the steady state)
            hs = [lib.Batch(cfg, tasks, device=(0 if EMU else local_rank)) for _ in range(W)]
            for h in hs:
                h.set_output(abi.OUT_EXECUTE)

            def passes(n_each):
                def body(h):
                    set_dev(torch, local_rank)
                    for _ in range(n_each):
                        h.run_pass(); h.fetch_raw(1)     # one pass = snf_batch_pass (call_candidates + finalize; replayed as a graph for small batches), as the headline runs it
                ths = [threading.Thread(target=body, args=(h,)) for h in hs]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            passes(warm)
            dev_sync(torch)
            t0 = time.perf_counter()
            passes(steps // W)
            dev_sync(torch)
            dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            hs[0].run_pass(); n_ret = hs[0].fetch_raw(1)
            lat = (time.perf_counter() - t1) * 1e3
            hs[0].set_output(abi.OUT_CANDIDATES); hs[0].call_candidates(); hs[0].finalize(); got = hs[0].fetch(1)
            hs[0].set_output(abi.OUT_EXECUTE); hs[0].call_candidates(); hs[0].finalize(); exe = hs[0].fetch(1)
            base, ver = cpu_baseline_and_verify(a, wl, got, [ci for ci, _ in specs], exe, cfg)
            n_sig = sum(t.n_leads for t in tasks)
            out[str(k)] = dict(workload=wl["name"], signatures=n_sig, steps=steps // W * W, batches_in_flight=W,
                               ms_per_step=round(dt / (steps // W * W) * 1e3, 3), ms_one_batch_in_flight=round(lat, 3),
                               signatures_per_s=round(n_sig * (steps // W * W) / dt), candidates=int(len(got.calls)), records_returned=int(n_ret),
                               verified=ver["ok"], differences=ver["differences"], cpu_all_core_sig_s=round(base["all_core_sig_s"]),
                               cpu_cores=base["cores"])
            for h in hs:
                h.close()
            if not args.no_reference_baseline:       # ... and against the unmodified reference itself, on this box
                try:
                    rc = reference_check(a, wl, exe, tasks, specs)
                    if rc is not None:
                        out[str(k)].update(rc)
                        out[str(k)]["vs_reference_all_cores"] = round(out[str(k)]["signatures_per_s"] / max(1, rc["reference_all_core_sig_s"]), 1)
                except Exception as e:  # noqa: BLE001
                    out[str(k)]["reference_error"] = f"{type(e).__name__}: {str(e)[:300]}"
            out[str(k)]["seconds"] = round(time.time() - t_all, 1)
        except Exception as e:  # noqa: BLE001 - the headline must not die with a side measurement
            out[str(k)] = dict(error=f"{type(e).__name__}: {e}")
    try:
        t_all = time.time()
        from tools import bench_population
        a = copy.copy(args); a.config = 4; a.steps = 2; a.warmup = 1
        r = bench_population.run(dict(ctx, args=a))
        out["4"] = dict(workload=r["config"]["workload"], metric=r["metric"], candidates=r["config"]["candidates"], combined_calls=r["config"]["combined_calls"],
                        ms_per_step=round(r["ms_per_step"], 1), candidates_per_s=round(r["value"]), steps=r["steps"], verified=r.get("verified"),
                        kernel_ms=r["config"].get("rank0", {}).get("kernel_ms"), parity_unpinned=r["config"].get("parity_unpinned"),
                        seconds=round(time.time() - t_all, 1))
    except Exception as e:  # noqa: BLE001
        out["4"] = dict(error=f"{type(e).__name__}: {e}")
    return out


def execute_mode_differences(got, exe, cfg) -> list:
    """SNF_OUT_EXECUTE against its definition (parallel.py:265-271) applied to the candidate-mode result on the host."""
    import numpy as np
    diffs = []
    keep = []
    for t in range(len(got.task_status)):
        lo, hi = int(got.task_call_off[t]), int(got.task_call_off[t + 1])
        idx = np.arange(lo, hi)
        if not cfg.no_qc:
            idx = idx[got.calls["qc"][lo:hi] != 0]
        if cfg.sort:
            idx = idx[np.argsort(got.calls["pos"][idx], kind="stable")]
        keep.append(idx)
    idx = np.concatenate(keep) if keep else np.zeros(0, np.int64)
    if len(idx) != len(exe.calls) or exe.task_call_off.tolist() != np.concatenate([[0], np.cumsum([len(k) for k in keep])]).tolist():
        return [f"execute mode: {len(exe.calls)} records, expected {len(idx)}"]
    for f in exe.calls.dtype.names:
        if f in ("alt_off", "rn_off"):
            continue
        a, e = exe.calls[f], got.calls[f][idx]
        if not np.array_equal(a, e, equal_nan=a.dtype.kind == "f"):
            diffs.append(f"execute mode: field {f} differs")

    def gather(pool, off, ln):
        ln = np.maximum(ln.astype(np.int64), 0)
        first = np.cumsum(ln) - ln
        return pool[np.repeat(off.astype(np.int64) - first, ln) + np.arange(int(ln.sum()), dtype=np.int64)]
    if not np.array_equal(gather(exe.alt_pool, exe.calls["alt_off"], exe.calls["alt_len"]), gather(got.alt_pool, got.calls["alt_off"][idx], got.calls["alt_len"][idx])):
        diffs.append("execute mode: ALT bytes differ")
    if not np.array_equal(gather(exe.rnames, exe.calls["rn_off"], exe.calls["rn_len"]), gather(got.rnames, got.calls["rn_off"][idx], got.calls["rn_len"][idx])):
        diffs.append("execute mode: read names differ")
    return diffs


def cpu_baseline_and_verify(args, wl, got, task_keys, exe=None, cfg=None):
    """The C oracle (scalar restatement of the reference, oracle/snf_oracle.c) over the WHOLE workload on this box's host
    cores: one process per contig task, at most one per core (the reference's schedule, `sniffles:495-530`).  A reported
    baseline, not the target.  With `got` (the HIP results of the bench batch) the same run is the checker of --verify."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_pool
    from sniffles_amd import records
    specs = task_specs(args, wl, 0, 0, 1)
    r = cpu_pool.run_tasks(specs, wl["cfg"], weights=[kw["contig_len"] for _, kw in specs], want_results=got is not None)
    n = sum(m["n_leads"] for m in r["items"].values())
    base = dict(value=n / r["hot_all_core_s"], unit="signatures/s", cores=r["procs"], kind="port",
                cores_used=r["procs"], host_cores=r["cores"],
                all_core_sig_s=n / r["hot_all_core_s"], single_core_sig_s=n / r["hot_single_core_s"],
                reference_cpython_sig_s=REFERENCE_CPYTHON["sig_s"], reference_cpython_host=REFERENCE_CPYTHON["host"],
                sample=f"the whole workload ({len(specs)} contig tasks, {n} signatures), one oracle process per contig "
                       f"({r['procs']} processes, longest contig first), call_candidates + finalize_candidates only: slowest "
                       f"process {r['hot_all_core_s']:.3f} s, sum over tasks {r['hot_single_core_s']:.3f} s "
                       f"(wall incl. the dense coverage vector each task builds first: {r['wall_s']:.2f} s)")
    ver = None
    if got is not None:
        diffs, n_calls = [], 0
        contig_of = {key: kw["contig"] for key, kw in specs}
        for t, key in enumerate(task_keys):     # task t of the bench batch is the contig with this key
            exp = r["items"][key]["result"]
            n_calls += int(exp.calls.shape[0])
            for d in records.diff_results(got, t, exp, 0):
                diffs.append(f"task {t} ({contig_of[key]}): {d}")
        if exe is not None:
            diffs += execute_mode_differences(got, exe, cfg)
        ver = dict(ok=not diffs, tasks=len(specs), calls_compared=n_calls, records_returned=(int(len(exe.calls)) if exe is not None else None),
                   what="every field of every candidate record, ALT bytes, supporting reads and coverage_average_total of the "
                        "bench batch vs the C oracle on the same inputs; the block the timed passes return (--output execute) vs "
                        "CallTask.execute's filter + sort applied to those candidates", differences=diffs[:5])
    return base, ver


def reference_differences(r, exe, tasks, task_keys):
    """The execute-mode block `exe` against what the unmodified reference's CallTask.execute keeps (the records a ref_pool run `r`
    returned), record by record and field by field: (differences, records compared)."""
    from sniffles_amd import records
    got = records.records(exe, tasks, "final")
    diffs, n_cmp = [], 0
    for t, key in enumerate(task_keys):
        exp = r["items"][key]["records"]
        g = got[t]
        n_cmp += len(exp)
        if isinstance(g, dict) or len(g) != len(exp):
            diffs.append(f"task {t}: {len(exp)} reference records, got {g if isinstance(g, dict) else len(g)}")
            continue
        for a, b in zip(g, exp):
            if a != b:
                diffs.append(f"task {t} {b['id']}: " + ", ".join(k for k in b if a.get(k) != b.get(k)))
                if len(diffs) > 5:
                    break
    return diffs, n_cmp


def reference_check(a, wl, exe, tasks, specs):
    """A side configuration against the LIVE reference on this box (oracle/_ref through oracle/ref_pool.py): the reference's rate on the
    host cores and the record-by-record comparison of the execute-mode block.  None where the staged reference is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_pool
    if not ref_pool.available():
        return None
    extra = ["--mosaic"] if wl["cfg"].get("mosaic") else []
    r = ref_pool.run_tasks(specs, extra, weights=[kw["contig_len"] for _, kw in specs], want_results=True,
                           max_procs=int(os.environ.get("SNF_BENCH_REF_PROCS", "0")) or None)
    diffs, n_cmp = reference_differences(r, exe, tasks, [ci for ci, _ in specs])
    n = sum(m["n_leads"] for m in r["items"].values())
    return dict(verified_vs_reference=not diffs, records_compared=n_cmp, differences=diffs[:5], reference_all_core_sig_s=round(n / r["hot_all_core_s"]),
                reference_hot_all_core_s=round(r["hot_all_core_s"], 3), reference_procs=r["procs"], reference_leg_s=round(r["total_wall_s"], 1))


def reference_baseline(args, wl, exe, tasks, task_keys, out):
    """`cpu_baseline` with kind = "reference" (SURVEY.md 8d): the UNMODIFIED reference's `Task.call_candidates` +
    `finalize_candidates` (`parallel.py:104-201`) on the same 24 signature tables, one OS process per contig task, pool =
    min(tasks, host cores) - its own schedule (`sniffles:495-530`).  The reference is the byte-compiled staged build `oracle/_ref`
    (or the checkout in the build container).  With `exe` (the execute-mode block of the timed passes) the same run checks
    the GPU's records against the reference ITSELF: what `CallTask.execute` sends to the parent, record by record."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_pool
    if not ref_pool.available():
        return None
    from sniffles_amd import records
    specs = task_specs(args, wl, 0, 0, 1)
    extra = ["--mosaic"] if wl["cfg"].get("mosaic") else []
    r = ref_pool.run_tasks(specs, extra, weights=[kw["contig_len"] for _, kw in specs], want_results=exe is not None,
                           max_procs=int(os.environ.get("SNF_BENCH_REF_PROCS", "0")) or None)
    n = sum(m["n_leads"] for m in r["items"].values())
    ref_sig_s = n / r["hot_all_core_s"]
    base = dict(value=ref_sig_s, unit="signatures/s", cores=r["procs"], kind="reference", host_cores=r["cores"],
                all_core_sig_s=ref_sig_s, single_core_sig_s=n / r["hot_single_core_s"],
                hot_all_core_s=round(r["hot_all_core_s"], 3), hot_single_core_s=round(r["hot_single_core_s"], 2),
                reference=ref_pool.kind(),
                sample=f"the whole workload ({len(specs)} contig tasks, {n} signatures): the unmodified reference's Task.call_candidates + "
                       f"finalize_candidates, one process per contig ({r['procs']} processes on {r['cores']} usable cores, longest contig first; "
                       f"the reference cannot use more processes than contigs), every process starts at a barrier once its Lead tables / "
                       f"coverage vector are built (untimed: {r['build_single_core_s']:.0f} core-seconds of record_lead / record_hap_ref): "
                       f"slowest process {r['hot_all_core_s']:.2f} s, sum over tasks {r['hot_single_core_s']:.1f} s; whole leg {r['total_wall_s']:.0f} s")
    # speed-ups against the reference on THIS box (north_star: >= 20x wall clock at 1 MI355X vs all host cores)
    vs = dict(gpu_pass=round(out["value"] / ref_sig_s, 1))
    wc = out.get("wall_clock") or {}
    if wc.get("batched"):
        vs["wall_clock_batched"] = round(r["hot_all_core_s"] * 1e3 / wc["batched"]["end_to_end_ms"], 1)
        vs["wall_clock_per_task_api"] = round(r["hot_all_core_s"] * 1e3 / wc["per_task_api"]["end_to_end_ms"], 1)
        if wc.get("per_task_execute"):
            vs["wall_clock_per_task_execute"] = round(r["hot_all_core_s"] * 1e3 / wc["per_task_execute"]["end_to_end_ms"], 1)
        if isinstance(wc.get("worker_processes"), dict):
            vs["wall_clock_worker_processes"] = {k: round(r["hot_all_core_s"] * 1e3 / m["hot_all_ms"], 1)
                                                 for k, m in wc["worker_processes"].items() if isinstance(m, dict) and m.get("hot_all_ms")}
    vs["note"] = ("reference all-core seconds for one genome / this package's seconds for one genome: gpu_pass = the timed step (inputs in HBM, "
                  "result block on the host); wall_clock_batched = numpy columns -> upload -> pass -> SVCall objects; per_task_api = 24 x "
                  "Task.call_candidates / finalize_candidates")
    base["vs_baseline"] = vs
    if exe is not None:
        diffs, n_cmp = reference_differences(r, exe, tasks, task_keys)
        base["verified_vs_reference"] = dict(ok=not diffs, records_compared=n_cmp, differences=diffs[:5],
                                             what="the execute-mode block of the timed passes (every field: POS, END, SVLEN, SVTYPE, support, GT/GQ/DR/DV, "
                                                  "filters, fp64 statistics, INS consensus ALT, supporting read names) vs what the unmodified reference's "
                                                  "CallTask.execute keeps (parallel.py:265-271) on the same signature tables, on this box")
        out["verified_vs_reference"] = not diffs
    return base


if __name__ == "__main__":
    main()
