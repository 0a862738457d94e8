#!/usr/bin/env python3
"""Host phases of the population merge (`candstore.execute_many`, bench.py --config 4) on a box WITHOUT a GPU: the population is
built through the test tier's host emulation (tests/emu), the group assignment (`snf_combine_resolve_batch`) runs there ONCE and
its output is replayed for the timed passes - what is measured is everything around that call (column assembly, sort and flush
windows, membership, emission order, the text of the merged records), which is host code and the same on any box.
Development tool: prints the phases of `candstore.last_timing`; with `--cprofile` the Python-level profile of one pass.
`--gpu`: on a box with a GPU - the library itself, nothing replayed: the merge as bench.py --config 4 times it, for every number of
runs of tasks in `--chunks` (SNF_COMBINE_CHUNKS) on ONE population.

    python tools/combine_host_prof.py [--scale 0.2] [--samples 10] [--passes 5] [--cprofile]
    python tools/combine_host_prof.py --gpu --scale 1.0 --chunks 1,2,3,4,6"""
import argparse
import ctypes
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GPU = "--gpu" in sys.argv
if not GPU:
    os.environ["SNF_BENCH_EMU"] = "1"
    from tools.bench_common import emu_lib  # noqa: E402
    emu_lib()
import numpy as np  # noqa: E402

from sniffles_amd import candstore, lib, parallel, synth, vcf  # noqa: E402
from sniffles_amd.config import SnifflesConfig  # noqa: E402
from tools.bench_population import build_sample  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--coverage", type=float, default=15.0)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--cprofile", action="store_true")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--chunks", default=None, help="comma-separated SNF_COMBINE_CHUNKS values to time one after the other")
    ap.add_argument("--sweep", default=None, help="environment settings to time one after the other on the one population, e.g. "
                    "SNF_COMBINE_WAVES=8+SNF_COMBINE_HEAVY=1e7,SNF_COMBINE_WAVES=1 (the library reads them at every call)")
    ap.add_argument("--cache", default=None, help="pickle of the emulated population and group assignment (built when absent)")
    a = ap.parse_args()
    contigs = [(ci, c, max(200000, int(synth.GRCH38[c] * a.scale))) for ci, c in enumerate(synth.CONTIGS)]
    call_cfg = SnifflesConfig()
    t0 = time.time()
    import pickle
    saved = {}
    if a.cache and os.path.exists(a.cache):
        with open(a.cache, "rb") as f:
            key, readers, n_cands, saved = pickle.load(f)
        assert key == (a.scale, a.samples, a.coverage), "the cache holds another population"
        print(f"population: {a.samples} samples, {n_cands} candidates, from {a.cache}", flush=True)
    else:
        readers, n_cands = {}, 0
        for s in range(a.samples):
            tasks = [synth.gen_task(ci, c, L, a.coverage, seed=100 + s, site_seed=501) for ci, c, L in contigs]
            readers[s], n = build_sample(call_cfg, tasks, 0, s)
            n_cands += n
        print(f"population: {a.samples} samples, {n_cands} candidates, built in {time.time() - t0:.1f} s (emulated)", flush=True)
    import gc
    gc.collect(); gc.freeze()
    cfg = SnifflesConfig()
    cfg.mode = "combine"
    cfg.snf_input_info = [dict(internal_id=s, sample_id=f"S{s}") for s in range(a.samples)]
    cfg.sample_ids_vcf = [(s, f"S{s}") for s in range(a.samples)]

    real_call = lib.combine_resolve_batch

    def replayed(config, problems, device=0):
        n = sum(int(p.n_cands) for p in problems)
        out = ctypes.cast(min(ctypes.cast(p.out_group, ctypes.c_void_p).value for p in problems), ctypes.POINTER(ctypes.c_int32))      # (the problems come in any order)
        if n not in saved:
            t = time.time()
            real_call(config, problems, device=device)
            saved[n] = np.ctypeslib.as_array(out, (n,)).copy()
            print(f"group assignment of {n} candidates emulated once in {time.time() - t:.1f} s; replayed from here on", flush=True)
        else:
            ctypes.memmove(out, saved[n].ctypes.data, 4 * n)
    if not GPU:
        lib.combine_resolve_batch = replayed

    text = [None]

    def one_pass():
        tasks = [parallel.CombineTask(id=ci, sv_id=0, contig=c, start=0, end=L - 1, config=cfg, device=0) for ci, c, L in contigs]
        buf = io.TextIOWrapper(io.BytesIO(), encoding="utf-8", newline="", write_through=True)      # (as open(path, "w") gives the writer)
        w = vcf.VCF(cfg, buf)
        n = sum(w.write_merged(part) for part in parallel.CombineTask.execute_many(tasks, readers, text_writer=w))
        buf.flush()
        text[0] = buf
        return n

    one_pass()
    first = text[0].buffer.getvalue().decode("utf-8")
    if a.cache and not os.path.exists(a.cache):
        for r in readers.values():
            r.__dict__.pop("_snf_columns", None)
        with open(a.cache, "wb") as f:
            pickle.dump(((a.scale, a.samples, a.coverage), readers, n_cands, saved), f, protocol=4)
        one_pass()
    import hashlib
    print("text sha1", hashlib.sha1(first.encode()).hexdigest(), flush=True)
    settings = [("SNF_COMBINE_CHUNKS=" + c) for c in a.chunks.split(",")] if a.chunks else a.sweep.split(",") if a.sweep else [None]
    touched = set()
    for chunks in settings:
        if chunks is not None:
            for name in touched:
                os.environ.pop(name, None)
            for kv in chunks.split("+"):
                name, val = kv.split("=", 1)
                os.environ[name] = val; touched.add(name)
            one_pass()
        best, best_all = None, None
        for k in range(a.passes):
            t = time.perf_counter()
            n = one_pass()
            dt = (time.perf_counter() - t) * 1e3
            ph = {k_: (round(v * 1e3, 2) if isinstance(v, float) else v) for k_, v in candstore.last_timing.items()}
            host = dt - ph.get("resolve_groups_gpu", 0.0)
            print(f"chunks {chunks} pass {k}: {dt:.1f} ms" + ("" if GPU else f", without the replayed call {host:.1f} ms") +
                  f", {n} records, text {text[0].buffer.tell()} B  {ph}", flush=True)
            best = host if best is None else min(best, host)
            best_all = dt if best_all is None else min(best_all, dt)
        assert text[0].buffer.getvalue().decode("utf-8") == first
        print(f"chunks {chunks}: best pass {best_all:.1f} ms" + ("" if GPU else f", best host time around the call {best:.1f} ms"), flush=True)
    if a.cprofile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable(); one_pass(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
