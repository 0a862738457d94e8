#!/usr/bin/env python3
"""Benchmark of the MI355X-native Sniffles2 hot path (BASELINE.json metric).

One "step" = one full pass of the hot path (Task.call_candidates + Task.finalize_candidates of every
contig task of the workload: binning, clustering, candidate calls, coverage, QC, genotyping, phasing,
INS consensus) with the signature tables already resident in HBM, INCLUDING the device->host copy of
the call records / ALT pool and, for N > 1, the RCCL gather of the per-rank call records on rank 0.

Workload at N = 1: BASELINE.json configs[1], "30x ONT HG002 whole-genome germline" restated as a seeded
synthetic signature set (24 GRCh38 contigs).  N > 1: weak scaling - N genome replicas
(seed 1..N), the 24*N contig tasks sharded longest-first over the ranks (contigs are independent,
SURVEY.md 8e); the only collective is the gather of the call records on rank 0 (counts first, then the records).

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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CPU_SAMPLE_CONTIGS = ["chr16", "chr17", "chr18", "chr19", "chr20", "chr21", "chr22"]


def shard_tasks(n_ranks: int):
    """(replica, contig) tasks, longest-processing-time-first over ranks.  Deterministic, no communication."""
    from sniffles_amd import dist as sdist, synth
    items = [(rep, ci, c) for rep in range(n_ranks) for ci, c in enumerate(synth.CONTIGS)]
    shards = sdist.shard_lpt([synth.GRCH38[c] for _, _, c in items], n_ranks)
    return [[items[i] for i in s] for s in shards]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every contig (debug only; invalid as a result)")
    ap.add_argument("--genomes", type=int, default=1,
                    help="genome replicas per batch and rank (SURVEY.md 8d scale knob; the headline configuration is 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight per GPU (host threads, each with its own batch handle and streams): the "
                         "device->host copies, host waits and launch-bound phases of one pass overlap the kernels of "
                         "the other.  1 = strictly one pass at a time")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the result): libraries that print banners through C stdio (RCCL prints its version
    # block to stdout, flushed only at exit when stdout is a pipe) are sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # SNF_BENCH_FORCE_DIST=1: run the whole collective path (RCCL process group, gathers from the worker threads) with a
    # single rank - a dry run of the N > 1 code on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SNF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from sniffles_amd import abi, lib, synth
    from sniffles_amd.config import SnifflesConfig

    cfg = SnifflesConfig()  # germline defaults (config.py)
    my = shard_tasks(world)[rank]
    t0 = time.time()
    tasks = []
    for g in range(max(1, args.genomes)):
        for k, (rep, ci, c) in enumerate(my):
            L = max(200000, int(synth.GRCH38[c] * args.scale))
            ti = synth.gen_task((g * world + rep) * 24 + ci, c, L, args.coverage, seed=1 + rep + 1000 * g)
            tasks.append(ti)
    n_sig = sum(t.n_leads for t in tasks)
    n_reads = sum(t.n_reads for t in tasks)
    seq_bytes = sum(int(t.seq_pool.nbytes) for t in tasks)
    t_gen = time.time() - t0

    import threading
    W = max(1, args.inflight)
    t0 = time.time()
    batches = [lib.Batch(cfg, tasks, device=local_rank) for _ in range(W)]  # same input, W independent handles
    t_upload = time.time() - t0

    cap_t = torch.tensor([max(1024, n_sig // 8)], dtype=torch.int64, device="cuda")
    if use_dist:
        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)  # one capacity on every rank
    cap_calls = int(cap_t.item())
    rec_bytes = abi.CALL_DTYPE.itemsize
    if use_dist:
        sends = [torch.empty(cap_calls * rec_bytes, dtype=torch.uint8, device="cuda") for _ in range(W)]
        count_t = torch.zeros(1, dtype=torch.int64, device="cuda")
        # the records are gathered on rank 0 only (one gather at the end; the parent writes the output)
        gathered = torch.empty(world * cap_calls * rec_bytes, dtype=torch.uint8, device="cuda") if rank == 0 else None
        counts = torch.zeros(world, dtype=torch.int64, device="cuda")
        counts_h = torch.zeros(world, dtype=torch.int64).pin_memory()
    # Collectives run on ONE communication thread per rank, in the order the passes finish: a worker thread exports its
    # records (device-to-device) into the handle's send buffer, queues the handle and goes on with its next pass; the
    # gather overlaps that pass.  Every rank issues the same (counts, records) sequence, so the order matches everywhere.
    import queue
    comm_q = queue.Queue()
    send_free = [threading.Event() for _ in range(W)]
    for ev in send_free:
        ev.set()
    comm_err = []

    def comm_loop():
        try:
            torch.cuda.set_device(local_rank)
            while True:
                item = comm_q.get()
                if item is None:
                    comm_q.task_done()
                    return
                w, nexp = item
                count_t.fill_(nexp)
                dist.all_gather_into_tensor(counts, count_t)          # 8 bytes per rank
                counts_h.copy_(counts, non_blocking=True)
                torch.cuda.current_stream().synchronize()              # (releases the GIL while it waits)
                nmax = int(counts_h.max()) * rec_bytes                 # every rank sends the same, smallest sufficient size
                chunk = sends[w][:nmax]
                if rank == 0:
                    dist.gather(chunk, gather_list=[gathered[r * nmax:(r + 1) * nmax] for r in range(world)], dst=0)
                else:
                    dist.gather(chunk, dst=0)
                torch.cuda.current_stream().synchronize()              # the send buffer may be reused
                send_free[w].set()
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

    def step(w):
        """One full pass of the hot path over batch w: candidates, finalize, D2H of the results, gather."""
        batch = batches[w]
        t_a = time.perf_counter()
        batch.call_candidates()
        t_b = time.perf_counter()
        batch.finalize()
        t_c = time.perf_counter()
        n = batch.fetch_raw(1)  # call records + ALT pool + read names on the host (blocks)
        t_d = time.perf_counter()
        if w == 0:
            phase_s[0] += t_b - t_a; phase_s[1] += t_c - t_b; phase_s[2] += t_d - t_c; phase_s[3] += 1
        if use_dist:
            send_free[w].wait()                                        # the previous gather of this handle has left the buffer
            send_free[w].clear()
            nexp = batch.export_calls_device(sends[w].data_ptr(), cap_calls)
            batch.sync()                                               # the copy is on the handle's stream
            comm_q.put((w, nexp))
        return n

    def barrier():
        if use_dist:
            comm_q.join()                                              # every queued gather has completed
            if comm_err:
                raise comm_err[0]
            dist.barrier()
        torch.cuda.synchronize()

    n_calls_box = [0]

    def run_passes(total):
        """Exactly `total` passes, split evenly over the W host threads (thread w works on its own batch handle)."""
        if W == 1:
            for _ in range(total):
                n_calls_box[0] = step(0)
            return
        errs = []

        def worker(w, k):
            try:
                torch.cuda.set_device(local_rank)
                for _ in range(k):
                    n_calls_box[0] = step(w)
            except BaseException as e:  # noqa: BLE001 - re-raised in the main thread
                errs.append(e)

        ths = [threading.Thread(target=worker, args=(w, total // W + (1 if w < total % W else 0))) for w in range(W)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    run_passes(-(-args.warmup // W) * W)  # >= warmup passes, the same number on every handle
    barrier()
    t0 = time.perf_counter()
    run_passes(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    n_calls = n_calls_box[0]
    timings = batches[0].timings()  # per-kernel HIP-event times of handle 0's LAST pass in the timed region, on its streams
    # reference point outside the timed region: the same pass with ONE batch in flight (per-pass latency)
    lat_ms = None
    if W > 1:
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            step(0)
        torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / 5 * 1e3
    barrier()   # also drains the communication thread before the main thread issues collectives again
    timings_alone = dict((k[0], k[1]) for k in batches[0].timings()) if lat_ms else {}

    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    tot = torch.tensor([n_sig, n_calls], dtype=torch.int64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    total_sig, total_calls = int(tot[0].item()), int(tot[1].item())

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = total_sig * args.steps / dt_max
        # dominant kernel of the last step, measured with HIP events around each launch
        kern = sorted(timings, key=lambda x: -x[1])
        # dominant KERNEL: entries that bracket a sequence of library launches (rocPRIM sort / scan passes) or a copy are
        # listed in top_kernels but are not a kernel whose roofline could be stated
        single = [k for k in kern if not k[0].startswith(("sort_", "scan_", "d2h_"))]
        top = single[0] if single else ("none", 0.0, 0)
        gpu_ms = sum(k[1] for k in kern)
        achieved = (top[2] / (top[1] * 1e-3)) / 1e9 if top[1] > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC counters can not be read from inside the process; they were collected
        # with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes) on this same workload and are kept, corrected as
        # the microarch guide prescribes, in profiles/r01_pmc_traffic.json
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            if top[0] in pmc and args.scale == 1.0 and args.coverage == 30.0:
                traffic = pmc[top[0]]["hbm_bytes"]
        except Exception:
            traffic = None
        roofline = dict(bound="hbm", kernel=top[0], achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                        kernel_ms=round(top[1], 4), algorithmic_bytes=int(top[2]),
                        gpu_ms_all_kernels=round(gpu_ms, 3),
                        top_kernels=[dict(name=k[0], ms=round(k[1], 4), algorithmic_bytes=int(k[2]),
                                          **({"ms_one_batch_in_flight": round(timings_alone[k[0]], 4)} if k[0] in timings_alone else {}))
                                     for k in kern[:8]])
        out = dict(metric="SV-signatures clustered/sec (clustering + calling + QC + genotype + INS consensus)",
                   value=value, unit="signatures/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int32/f64",
                   data="synthetic",
                   config=dict(workload="30x ONT HG002-shaped whole-genome germline, 24 GRCh38 contigs per replica "
                                        "(BASELINE.json configs[1]), synthetic signature tables (SURVEY.md 8d)",
                               replicas=world, genomes_per_batch=max(1, args.genomes), tasks=24 * world * max(1, args.genomes), coverage=args.coverage, scale=args.scale,
                               signatures=total_sig, reads_rank0=n_reads, ins_seq_bytes_rank0=seq_bytes,
                               calls=total_calls, parallelism=f"contig-sharded x{world}, RCCL gather of the call records on rank 0",
                               batches_in_flight_per_gpu=W,
                               ms_per_pass_one_batch_in_flight=(round(lat_ms, 3) if lat_ms else None),
                               gen_s=round(t_gen, 2), upload_s=round(t_upload, 2),
                               host_ms_per_step=dict(enqueue_call_candidates=round(phase_s[0] / phase_s[3] * 1e3, 3),
                                                     finalize_incl_2_syncs=round(phase_s[1] / phase_s[3] * 1e3, 3),
                                                     fetch_d2h=round(phase_s[2] / phase_s[3] * 1e3, 3))),
                   roofline=roofline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm_thread is not None:
        comm_q.put(None)
        comm_thread.join(timeout=30)
    for bb in batches:
        bb.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(cfg, args):
    """Oracle (scalar C restatement of the reference, oracle/snf_oracle.c) timed on this box's host cores on a
    bounded sample of the same workload.  A reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from sniffles_amd import synth
    oracle.build()
    tis = [synth.gen_task(synth.CONTIGS.index(c), c, max(200000, int(synth.GRCH38[c] * args.scale)), args.coverage, seed=1)
           for c in CPU_SAMPLE_CONTIGS]
    n = sum(t.n_leads for t in tis)
    t0 = time.perf_counter()
    oracle.run(cfg, tis, True)
    wall = time.perf_counter() - t0
    hot = oracle.hot_seconds()
    return dict(value=n / hot, unit="signatures/s", cores=1, kind="port",
                sample=f"{'+'.join(CPU_SAMPLE_CONTIGS)} of replica 0 ({n} signatures), call_candidates+finalize only: "
                       f"{hot:.2f}s (wall incl. dense coverage build {wall:.2f}s); host cores available: {os.cpu_count()}")


if __name__ == "__main__":
    main()
