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

from tools.bench_common import DEV, EMU, WORKLOADS, dev_sync, emu_lib, set_dev, task_specs  # noqa: E402,F401
from tools.bench_legs import (bind_to_gpu_numa, cpu_baseline_and_verify, other_configs, reference_baseline,  # noqa: E402,F401
                              reference_check, wall_clock, worker_processes)
from tools.bench_roofline import HBM_PEAK_GBS, committed_issue, committed_profiles, pcie_d2h_peak_gbs  # noqa: E402,F401


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
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: loopback (the box's hostname need not resolve)
        if EMU:
            dist.init_process_group(backend="gloo")
        elif os.environ.get("SNF_BENCH_PG") == "nccl":
            # (the process group of rounds 2-5: RCCL's communicator - its streams and proxy thread on the device - exists from the start)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            # control words (a few int64 per run and per pass: sizes, layouts, the timing reduction, barriers) travel as CPU tensors
            # over gloo; RCCL - backend "nccl" for CUDA tensors - builds its communicator with the first CUDA collective, i.e. only where
            # result BLOCKS are gathered (no /dev/shm: several nodes, SNF_BENCH_GATHER=rccl).  On one node no result byte needs it
            # (dist.SharedLanding), and a communicator that merely exists cost the passes 14-16 % (profiles/r05_shared_probe.log, r06_shared_probe.log)
            dist.init_process_group(backend="cpu:gloo,cuda:nccl")

    # where the control tensors of the collectives live: host memory unless the whole group is RCCL
    ctl_dev = DEV if (not use_dist or EMU or os.environ.get("SNF_BENCH_PG") == "nccl") else "cpu"
    ctx = dict(args=args, rank=rank, world=world, local_rank=local_rank, use_dist=use_dist, numa=numa, ctl_dev=ctl_dev)
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
                    out["strong"] = dict(value=out2["value"], unit=out2["unit"], ms_per_step=out2["ms_per_step"], steps=out2["steps"],
                                         scaling="strong",
                                         signatures=out2["config"]["signatures"], calls=out2["config"]["calls"],
                                         parallelism=out2["config"]["parallelism"],
                                         gathered_on_rank0=out2["config"]["gathered_on_rank0"], ranks_seen=out2.get("ranks_seen"),
                                         sets_served=out2.get("sets_served"),
                                         note="ONE genome over the ranks of this run (one LPT-balanced contig set per rank and pass, "
                                              "claimed from the "
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
    CTL = ctx.get("ctl_dev", DEV)
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
    out_mode = (abi.OUT_EXECUTE if args.output == "execute" else abi.OUT_CANDIDATES) | (abi.OUT_DEVICE if use_dist and not shared
                                                                                        and not strong_shared else 0)
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
            need_b = max(need_b, 256 + len(res0.calls) * abi.CALL_DTYPE.itemsize + 4 * len(res0.rnames))
            need_a = max(need_a, len(res0.alt_pool))
        need = torch.tensor([need_b, need_a], dtype=torch.int64, device=CTL)
        dist.all_reduce(need, op=dist.ReduceOp.MAX)
        # /dev/shm must hold every rank's segments (containers often cap it): otherwise the block gather over RCCL is taken
        n_slots = NGEN * W * len(group_tasks) if strong_shared else 2 * W
        seg_bytes = n_slots * (int(need[0]) * 3 // 2 + int(need[1]) * 3 // 2 + (2 << 20) + 8192)
        try:
            st = os.statvfs("/dev/shm")
            # (SNF_BENCH_SHM_FULL=1: the test of this fallback)
            room = st.f_bavail * st.f_frsize >= int(world * seg_bytes * 1.25) and os.environ.get("SNF_BENCH_SHM_FULL") != "1"
        except OSError:
            room = False
        ok_t = torch.tensor([1 if room else 0], dtype=torch.int64, device=CTL)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if not int(ok_t.item()):
            if rank == 0:
                print(f"[bench] /dev/shm cannot hold {world} x {seg_bytes >> 20} MiB of result segments: gathering the blocks over RCCL "
                      f"instead", file=sys.stderr)
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
        landing = sdist.SharedLanding(slots=n_slots, block_bytes=int(need[0]) * 3 // 2 + (1 << 20),
                                      alt_bytes=int(need[1]) * 3 // 2 + (1 << 20), group=meta_group)
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
        cap_t = torch.tensor([blk * (len(group_tasks) if strong else 1) * 3 // 2 + (1 << 20)], dtype=torch.int64, device=CTL)
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
                    dbg = os.environ.get("SNF_BENCH_SHARED_DEBUG", "")      # probes of this path only: "oneslot", "nogather", "noset"
                    slot = 2 * w + (0 if "oneslot" in dbg else (turn_box[w] & 1)); turn_box[w] += 1
                    if "nogather" not in dbg:
                        send_free[slot].wait()                         # the parent has indexed the result this segment held two passes ago
                        send_free[slot].clear()
                    if "noset" not in dbg or turn_box[w] <= 2:
                        handles_box[0][w][0].set_result_memory(*landing.memory(slot))
                    one_pass(w)
                    served_box[0] += 1
                    n_calls_box[0] = lay_box[w]["n_calls"]
                    if "nogather" not in dbg:
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
        # (the library stages results through HBM while passes overlapped within the last 0.1 s: this is the lone-pass reference point)
        time.sleep(0.25)
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

    tt = torch.tensor([dt], dtype=torch.float64, device=CTL)
    tot = torch.tensor([n_sig, n_calls], dtype=torch.int64, device=CTL)
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
        # (front_window brackets the launches of the window front end: a stage, below)
        single = [k for k in kern if not k[0].startswith(("sort_", "scan_", "d2h_", "front_"))]
        top = single[0] if single else ("none", 0.0, 0)
        # stages of the pass (several launches each) with SURVEY.md 8(d)'s bytes: the dominant STAGE stands next to the dominant kernel
        by_name = {k[0]: k for k in kern}
        stage_def = [("window front end (w1 .. w6t)", ["front_window"], 21 * n_sig),
                     ("refine: merge_inner / resplit (d1g + d1w)", ["d1g_refine8", "d1w_refine"], 36 * n_sig),
                     ("call_from (d2g + d2w)", ["d2g_call8", "d2w_call"], 32 * n_sig),
                     ("INS consensus (SMALL + LARGE + verbatim)", ["e45w_consensus_small", "e45w_consensus_large", "e4c_copy"], None),
                     ("coverage annotation (d4 + d5w)", ["d4_coverage", "d5w_covsum"], 8 * n_reads + 20 * n_calls)]
        stages = []
        for nm, parts, nbytes in stage_def:
            ms_ = sum(by_name[p][1] for p in parts if p in by_name)
            if ms_ <= 0:
                continue
            nb_ = nbytes if nbytes is not None else sum(by_name[p][2] for p in parts if p in by_name)
            stages.append(dict(stage=nm, ms=round(ms_, 4), algorithmic_bytes=int(nb_),
                               frac=round(nb_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)))
        stages.sort(key=lambda x: -x["ms"])
        issue = committed_issue()
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
        result_bytes = (len(res_one.calls) * abi.CALL_DTYPE.itemsize + 4 * len(res_one.rnames)
                        + len(res_one.alt_pool)) if res_one is not None else None
        pcie_peak = pcie_d2h_peak_gbs(torch)
        result_path = dict(bound="pcie",
                           what="bytes of the result block one pass hands to the host (records + read names + ALT bytes of the "
                           + ("calls CallTask.execute keeps" if args.output == "execute" else "candidates")
                           + "), stored by the kernels straight into pinned "
                           "host memory, over the time of a step; peak = device -> pinned host copy rate measured in this run",
                           bytes_per_pass=result_bytes, records_per_pass=(len(res_one.calls) if res_one is not None else None),
                           achieved=(round(result_bytes / (ms_per_step * 1e-3) / 1e9, 2) if result_bytes else None),
                           peak=round(pcie_peak, 2), unit="GB/s",
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
                        kernel_ms_source="mean HIP-event duration of the kernel's launches (events on the stream it is launched on) over "
                                         "the timed passes of handle 0 (batches in flight as configured); "
                                         "the LARGE consensus kernel is bracketed on every pass, the other kernels on every 8th pass of "
                                         "the handle (snf_batch_timing_every)",
                        rocprof_ms=(round(rocprof_ms, 4) if rocprof_ms else None),
                        rocprof_frac=(round(top[2] / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if rocprof_ms else None),
                        profile_note=prof_note, result_path=result_path,
                        dominant_stage=(stages[0] if stages else None), stages=stages,
                        # the kernels of this path are integer / branch work bound by instruction issue and dependent latency, not by bytes:
                        # next to the HBM fraction the share of a kernel's time its VALU instructions alone need (profiles/r06_sq_all.txt)
                        issue=[dict(kernel=k[0], **issue[k[0]]) for k in kern if k[0] in issue][:8] or None,
                        gpu_ms_all_kernels=round(gpu_ms, 3),
                        whole_pass=dict(algorithmic_bytes=int(pass_bytes),
                                        achieved=round(pass_bytes / (ms_per_step * 1e-3) / 1e9, 2) if not strong and world == 1 else None,
                                        frac=round(pass_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   5) if not strong and world == 1 else None),
                        top_kernels=[dict(name=k[0], ms=round(k[1], 4), algorithmic_bytes=int(k[2]),
                                          **({"rocprof_ms": round(prof_avg[k[0]], 4)} if same_workload and k[0] in prof_avg else {}),
                                          **({"ms_one_batch_in_flight": round(timings_alone[k[0]], 4)} if k[0] in timings_alone else {}))
                                     for k in kern[:int(os.environ.get("SNF_BENCH_TOPK", "8"))]])
        n_contigs = len(wl["contigs"] or synth.CONTIGS)
        out = dict(metric="SV-signatures clustered/sec (clustering + calling + QC + genotype + INS consensus)",
                   value=value, unit="signatures/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_per_step, higher_is_better=True, scaling=args.scaling, vs_baseline=None, dtype="int32/f64",
                   ranks_seen=ranks_seen, sets_served=sets_served,
                   data="synthetic"
                   + (" - SNF_BENCH_EMU=1: host emulation of the kernels over gloo, a test of the N > 1 plumbing, NOT a result" if EMU
                      else "")
                   + (" - SNF_BENCH_CFG overrides the workload's configuration: an ablation, NOT a result"
                      if os.environ.get("SNF_BENCH_CFG") else ""),
                   config=dict(workload=wl["name"] + ", synthetic signature tables (SURVEY.md 8d)", baseline_config=args.config,
                               replicas=1 if strong else world, genomes_per_batch=G,
                               tasks=n_contigs * (1 if strong else world) * G,
                               coverage=args.coverage if args.coverage is not None else wl["coverage"], scale=args.scale,
                               signatures=total_sig, reads_rank0=n_reads, ins_seq_bytes_rank0=seq_bytes,
                               calls=total_calls,
                               parallelism=(f"one genome, {len(group_tasks)} contig sets (LPT-balanced, one device batch each) claimed "
                                            f"from a shared work queue by {world} ranks x {W} host threads, passes pipelined"
                                            if strong else f"contig-sharded x{world}")
                                            + (", every rank's result stored into node-shared host memory by its own kernels, layouts "
                                               "gathered on rank 0 (dist.SharedLanding)" if (shared or strong_shared)
                                               else ", RCCL gather of the result blocks on rank 0" if use_dist
                                               else ", one process: no gather"),
                               batches_in_flight_per_gpu=W, host_binding=ctx.get("numa"),
                               gathered_on_rank0=(dict(ranks=world, records=int(len(gathered_box[0].calls)),
                                                       alt_bytes=int(len(gathered_box[0].alt_pool)),
                                                       read_names=int(len(gathered_box[0].rnames)),
                                                       tasks=int(len(gathered_box[0].task_ids)),
                                                       order="task id, then position (parallel.py:270-271, sniffles:544)")
                                                  if use_dist and gathered_box[0] is not None else None),
                               ms_per_pass_one_batch_in_flight=(round(lat_ms, 3) if lat_ms else None),
                               output=args.output
                               + (": what CallTask.execute returns (parallel.py:265-271) - QC-passing calls, per task sorted by position, "
                                                     "filtered / sorted / compacted on the device" if args.output == "execute"
                                                     else ": every candidate record"),
                               timed_region="call_candidates + finalize (both enqueue only) + the one host wait of the pass, after which "
                                            "the result block "
                                            "(records, read names, ALT bytes) is in pinned host memory; the read index (sorted read "
                                            "ends + hap prefix counts = the coverage vector / hap tables the reference builds during "
                                            "extraction, excluded from cpu_baseline as well) is built once at upload",
                               ms_per_step_with_read_index_rebuilt_every_pass=(round(ms_with_index, 3) if ms_with_index else None),
                               gen_s=round(t_gen, 2), upload_s=round(t_upload, 2),
                               host_ms_per_step=dict(enqueue_call_candidates=round(phase_s[0] / max(1, phase_s[3]) * 1e3, 3),
                                                     finalize=round(phase_s[1] / max(1, phase_s[3]) * 1e3, 3),
                                                     fetch_d2h=round(phase_s[2] / max(1, phase_s[3]) * 1e3, 3)),
                               parity_unpinned=["edit distance vs edlib itself (edlib absent; pinned to the exact Levenshtein DP)",
                                                "pysam stand-in of the extraction oracle (pinned by the reference's 17 "
                                                "known-answer reads)"]),
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
                        base["reference_note"] = ("the staged reference build (oracle/_ref, made by oracle/make_ref.py during build() "
                                                  "where "
                                                  "/root/reference exists) is not on this box: kind stays \"port\"")
                if ver is not None:
                    out["verified"] = ver["ok"]
                    out["verify"] = ver
            if args.config == 1 and not args.no_configs and args.scale == 1.0 and args.coverage is None:
                for hs in handles:           # (the headline's batches are done: with their streams alive the small config ran 0.53
                    for bb in hs:            #  instead of 0.35 ms per step - more streams than hardware queues)
                        bb.close()
                out["configs"] = other_configs(ctx)
            # (the driver's record keeps `config` whole but only the NAMES of the other extra keys: the outcome of the checks goes there too)
            out["config"]["verified"] = out.get("verified")
            out["config"]["verified_vs_reference"] = out.get("verified_vs_reference")
            if "configs" in out:
                out["config"]["configs_verified"] = {k: dict(verified=v.get("verified"), verified_vs_reference=(
                    v["verified_vs_reference"].get("ok") if isinstance(v.get("verified_vs_reference"), dict) else v.get("verified_vs_reference")),
                    records_compared=(v["verified_vs_reference"].get("records_compared") if isinstance(v.get("verified_vs_reference"), dict)
                                      else v.get("records_compared"))) for k, v in out["configs"].items()}
    if comm_thread is not None:
        comm_q.put(None)
        comm_thread.join(timeout=30)
    for hs in handles:
        for bb in hs:
            bb.close()
    return out


if __name__ == "__main__":
    main()
