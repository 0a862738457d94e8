"""ctypes binding of libsniffles_amd.so (include/sniffles_amd.h) - the only compute path.

There is no CPU fallback: if the library is not built, or no HIP device is present,
every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsniffles_amd.so")
_lib = None


class SnifflesAmdError(RuntimeError):
    pass


def bind(lib: C.CDLL) -> C.CDLL:
    """Declare the prototypes of include/sniffles_amd.h on a loaded library."""
    vp = C.c_void_p
    lib.snf_abi_version.restype = C.c_int
    lib.snf_last_error.restype = C.c_char_p
    lib.snf_device_count.restype = C.c_int
    lib.snf_batch_create.argtypes = [C.POINTER(abi.snf_config_t), C.c_int, C.POINTER(vp)]
    lib.snf_batch_add_task.argtypes = [vp, C.POINTER(abi.snf_task_input_t)]
    lib.snf_batch_upload.argtypes = [vp]
    lib.snf_batch_destroy.argtypes = [vp]
    lib.snf_batch_destroy.restype = None
    lib.snf_batch_call_candidates.argtypes = [vp]
    lib.snf_batch_finalize.argtypes = [vp]
    lib.snf_batch_fetch.argtypes = [vp, C.c_int, C.POINTER(abi.snf_result_t)]
    lib.snf_batch_sync.argtypes = [vp]
    lib.snf_genotype_batch.argtypes = [C.POINTER(abi.snf_config_t), C.c_int, C.POINTER(abi.snf_call_t), C.c_int64]
    lib.snf_genotype_batch.restype = C.c_int
    lib.snf_batch_fetch_clusters.argtypes = [vp, C.c_int, C.POINTER(abi.snf_clusters_t)]
    lib.snf_batch_fetch_clusters.restype = C.c_int
    lib.snf_batch_export_device.argtypes = [vp, vp, C.c_int64, C.POINTER(abi.snf_export_layout_t)]
    lib.snf_batch_set_output.argtypes = [vp, C.c_int]
    lib.snf_batch_set_result_memory.argtypes = [vp, vp, C.c_int64, vp, C.c_int64]
    lib.snf_trim_caches.argtypes = [C.c_int]
    lib.snf_batch_n_candidates.argtypes = [vp]
    lib.snf_batch_n_candidates.restype = C.c_int64
    lib.snf_trim_caches.restype = C.c_int64
    lib.snf_batch_block_coverage.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
    lib.snf_batch_block_coverage.restype = C.c_int
    lib.snf_batch_coverage_calls.argtypes = [vp, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    lib.snf_batch_coverage_calls.restype = C.c_int
    lib.snf_batch_timing_count.argtypes = [vp]
    lib.snf_batch_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    lib.snf_batch_timing_every.argtypes = [vp, C.c_int]
    lib.snf_batch_timing_mean_reset.argtypes = [vp]
    lib.snf_batch_timing_mean_count.argtypes = [vp]
    lib.snf_batch_timing_mean_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.snf_edit_distance_batch.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_uint8),
                                            C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int32)]
    lib.snf_edit_distance_batch_k.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_uint8),
                                              C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.snf_edit_distance_batch_k.restype = C.c_int
    lib.snf_combine_resolve_batch.argtypes = [C.POINTER(abi.snf_config_t), C.c_int, C.POINTER(abi.snf_combine_problem_t), C.c_int64]
    lib.snf_combine_resolve_batch.restype = C.c_int
    lib.snf_combine_call_groups.argtypes = [C.POINTER(abi.snf_group_call_config_t), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]
    lib.snf_combine_call_groups.restype = C.c_int
    lib.snf_combine_last_stats.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.snf_combine_last_stats.restype = C.c_int
    u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    lib.snf_consensus_batch.argtypes = [C.c_int, C.c_int, u8p, C.c_int64, C.c_int64, i64p, i32p, i32p, i32p, i64p, i64p, i32p, u8p, i64p]
    lib.snf_consensus_batch.restype = C.c_int
    lib.snf_extract_create.argtypes = [C.POINTER(abi.snf_extract_config_t), C.c_int, C.POINTER(vp)]
    lib.snf_extract_upload.argtypes = [vp, C.POINTER(abi.snf_extract_input_t)]
    lib.snf_extract_run.argtypes = [vp]
    lib.snf_extract_result.argtypes = [vp, C.POINTER(abi.snf_extract_result_t)]
    lib.snf_extract_result_meta.argtypes = [vp, C.POINTER(abi.snf_extract_result_t)]
    lib.snf_extract_device_view.argtypes = [vp, C.POINTER(abi.snf_task_input_t), C.POINTER(C.c_int)]
    lib.snf_batch_add_task_device.argtypes = [vp, vp, C.POINTER(abi.snf_task_input_t)]
    lib.snf_extract_destroy.argtypes = [vp]
    lib.snf_extract_destroy.restype = None
    lib.snf_extract_last_error.restype = C.c_char_p
    for f in ("snf_extract_create", "snf_extract_upload", "snf_extract_run", "snf_extract_result", "snf_extract_result_meta",
              "snf_extract_device_view", "snf_batch_add_task_device"):
        getattr(lib, f).restype = C.c_int
    lib.snf_batch_pass.argtypes = [vp]
    lib.snf_batch_open.argtypes = [C.POINTER(abi.snf_config_t), C.c_int, C.POINTER(abi.snf_task_input_t), C.c_int32, C.c_int, C.POINTER(vp)]
    for f in ("snf_batch_open", "snf_batch_pass", "snf_batch_create", "snf_batch_add_task", "snf_batch_upload", "snf_batch_call_candidates",
              "snf_batch_finalize", "snf_batch_fetch", "snf_batch_sync", "snf_batch_export_device", "snf_batch_set_output", "snf_batch_set_result_memory",
              "snf_batch_timing_count",
              "snf_batch_timing_get", "snf_edit_distance_batch"):
        getattr(lib, f).restype = C.c_int
    if lib.snf_abi_version() != abi.ABI_VERSION:
        raise SnifflesAmdError("libsniffles_amd.so ABI version mismatch - rebuild (python -m sniffles_amd.build)")
    return lib


_override = None


def use_library(handle) -> None:
    """Every entry point of this package goes through the library `load()` returns.  `use_library(h)` makes that an already bound
    library `h` - another build of the same sources: `bind(ctypes.CDLL(path))`, the programmatic form of `SNF_LIB_SO`; the test
    suite's host tier (the unchanged HIP sources on a fibre stand-in for the HIP runtime) comes in this way - `use_library(None)`
    the default again.  Objects keep the library they were created with."""
    global _override
    _override = handle


def load() -> C.CDLL:
    global _lib
    if _override is not None:
        return _override
    if _lib is None:
        so = os.environ.get("SNF_LIB_SO") or SO      # SNF_LIB_SO: another build of the same sources (A/B measurements)
        if not os.path.exists(so):
            raise SnifflesAmdError(f"{so} is missing: build it with `python -m sniffles_amd.build` "
                                   "(the hot path has no CPU fallback)")
        _lib = bind(C.CDLL(so))
    return _lib


def _check(lib, rc):
    if rc != 0:
        raise SnifflesAmdError(lib.snf_last_error().decode("utf-8", "replace"))


class Batch:
    """A set of contig tasks resident in HBM.  `tasks`: list of sniffles_amd.soa.TaskInput."""

    def __init__(self, cfg, tasks, device: int = 0):
        self.lib = load()
        self.tasks = list(tasks)
        self._h = C.c_void_p()
        import os
        import time
        prof = os.environ.get("SNF_PROF") is not None
        t0 = time.perf_counter()
        cs = cfg if isinstance(cfg, abi.snf_config_t) else abi.config_struct(cfg)      # (a struct built elsewhere: sniffles_amd.server)
        _check(self.lib, self.lib.snf_batch_create(C.byref(cs), device, C.byref(self._h)))
        t1 = time.perf_counter()
        try:
            from .soa import DeviceTaskInput
            keep = []       # the task arrays are borrowed by the library until snf_batch_upload returns
            for ti in self.tasks:
                if isinstance(ti, DeviceTaskInput):     # columns already in HBM (extraction): device-to-device hand-over
                    ts = abi.task_meta_struct(ti, keep)
                    _check(self.lib, self.lib.snf_batch_add_task_device(self._h, ti._extractor._h, C.byref(ts)))
                    continue
                ts = abi.task_struct(ti, keep)
                _check(self.lib, self.lib.snf_batch_add_task(self._h, C.byref(ts)))
            t2 = time.perf_counter()
            _check(self.lib, self.lib.snf_batch_upload(self._h))
            if prof:
                import sys
                t3 = time.perf_counter()
                print(f"[SNF_PROF] Batch(): create {1e3 * (t1 - t0):.1f} ms, add_task x{len(self.tasks)} {1e3 * (t2 - t1):.1f} ms, "
                      f"upload {1e3 * (t3 - t2):.1f} ms", file=sys.stderr)
        except Exception:
            self.close()
            raise

    RUN_NONE, RUN_CANDIDATES, RUN_PASS = 0, 1, 0x100      # include/sniffles_amd.h SNF_RUN_*

    @classmethod
    def open_in_background(cls, cfg, tasks, device: int = 0, run: int = 0) -> "PendingBatch":
        """`Batch(cfg, tasks, device)` and, with `run`, its first device work (RUN_CANDIDATES: `call_candidates()`; RUN_PASS | output mode:
        `set_output(mode)` + `run_pass()`), started on a helper thread: everything that needs the interpreter (the structs of the C-ABI) is
        built here, on the caller's thread; the helper makes ONE library call (`snf_batch_open`: create, add, upload, enqueue), during which
        the interpreter lock is released.  `.result()` waits and returns the batch (or raises what the call raised)."""
        import threading
        from .soa import DeviceTaskInput
        lib_ = load()
        tasks = list(tasks)
        if any(isinstance(t, DeviceTaskInput) for t in tasks):
            # (columns that already live in HBM are handed over device-to-device by the synchronous constructor; snf_batch_open takes
            # host columns only - pulling them back to re-upload them would defeat the hand-over)
            raise TypeError("open_in_background takes host task inputs; use Batch(cfg, tasks) for device-resident ones")
        cs = abi.config_struct(cfg)
        keep = []
        arr = (abi.snf_task_input_t * max(1, len(tasks)))()
        for i, ti in enumerate(tasks):
            arr[i] = abi.task_struct(ti, keep)
        pend = PendingBatch()

        def body():
            h = C.c_void_p()
            try:
                _check(lib_, lib_.snf_batch_open(C.byref(cs), device, arr, len(tasks), int(run), C.byref(h)))
                b = cls.__new__(cls)
                b.lib, b.tasks, b._h = lib_, tasks, h
                pend._batch = b
            except BaseException as e:  # noqa: BLE001 - re-raised by result()
                pend._err = e
            keep.clear()
        pend._thread = threading.Thread(target=body, daemon=True)
        pend._thread.start()
        return pend

    def close(self):
        if self._h:
            self.lib.snf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def call_candidates(self):
        _check(self.lib, self.lib.snf_batch_call_candidates(self._h))

    def finalize(self):
        _check(self.lib, self.lib.snf_batch_finalize(self._h))

    def run_pass(self):
        """`call_candidates()` + `finalize()` as one unit (`snf_batch_pass`): the same results; from the second pass of a handle on
        the pass is replayed from a HIP graph (one host call instead of ~40 launches)."""
        _check(self.lib, self.lib.snf_batch_pass(self._h))

    def sync(self):
        _check(self.lib, self.lib.snf_batch_sync(self._h))

    def fetch(self, stage: int, copy: bool = True) -> abi.Result:
        """copy=False: views of the library's result buffers (valid until the next call on this batch) instead of copies."""
        r = abi.snf_result_t()
        _check(self.lib, self.lib.snf_batch_fetch(self._h, stage, C.byref(r)))
        return abi.Result(r, copy)

    def n_candidates(self) -> int:
        """Candidate calls of the last call_candidates (all tasks), as of the last fetch / sync."""
        return int(self.lib.snf_batch_n_candidates(self._h))

    def fetch_raw(self, stage: int) -> int:
        """D2H of the results into library-owned host memory without materialising numpy copies; returns n_calls."""
        r = abi.snf_result_t()
        _check(self.lib, self.lib.snf_batch_fetch(self._h, stage, C.byref(r)))
        return int(r.n_calls)

    def fetch_layout(self) -> dict:
        """The one host wait of a pass (stage-1 fetch) and where the result lies in the memory given to `set_result_memory`:
        `block` holds the records at 0 and the read names at `off_rnames`, `alt` the ALT section at 0 (the layout keys of
        `export_device`, with `off_alt` / `bytes` left to the caller, who knows how the two arrays are placed)."""
        r = abi.snf_result_t()
        _check(self.lib, self.lib.snf_batch_fetch(self._h, 1, C.byref(r)))
        n = int(r.n_calls)
        return dict(n_calls=n, rnames_len=int(r.rnames_len), alt_pool_len=int(r.alt_pool_len),
                    off_rnames=(n * abi.CALL_DTYPE.itemsize + 255) & ~255)

    def fetch_clusters(self, stage: int) -> dict:
        """The clusters of the candidate stage (0 seeds, 1 after the merge scan, 2 as cluster.resolve yields them) as numpy
        columns: task_index, svtype, start, end, seed, seed_index, n_leads_long, repeat, lead_off, lead (row in the task's
        input table), lead_svlen.  Needs call_candidates first."""
        r = abi.snf_clusters_t()
        _check(self.lib, self.lib.snf_batch_fetch_clusters(self._h, stage, C.byref(r)))
        n, m = int(r.n_clusters), int(r.n_leads)

        def col(p, k, dt):
            return np.ctypeslib.as_array(p, shape=(k,)).astype(dt, copy=True) if k else np.zeros(0, dt)
        out = {f: col(getattr(r, f), n, np.int32) for f in ("task_index", "svtype", "start", "end", "seed", "seed_index", "n_leads_long")}
        out["repeat"] = col(r.repeat, n, np.uint8)
        out["lead_off"] = col(r.lead_off, n + 1, np.int64)
        out["lead"], out["lead_svlen"] = col(r.lead, m, np.int32), col(r.lead_svlen, m, np.int32)
        return out

    def set_output(self, mode: int) -> None:
        """What a stage-1 fetch returns (before `finalize`): abi.OUT_CANDIDATES (default: every candidate, candidate order - what
        Task.finalize_candidates returns), abi.OUT_EXECUTE (what CallTask.execute keeps: qc-passing calls, per task sorted by
        pos, compacted on the device), optionally | abi.OUT_DEVICE (the block stays in HBM: `export_device`)."""
        _check(self.lib, self.lib.snf_batch_set_output(self._h, int(mode)))

    def set_result_memory(self, block=None, alt=None) -> None:
        """Where stage-1 results land on the host: two writable uint8 numpy arrays (e.g. views of a shared-memory segment another
        process maps as well) receive [records | read names] and the ALT section, written by the kernels themselves; the library
        page-locks them for the device (once per array: switching between a few segments is cheap).  None, None: the library's
        own pinned buffers again.  The arrays must outlive the batch; call it between passes."""
        if block is None or alt is None:
            _check(self.lib, self.lib.snf_batch_set_result_memory(self._h, None, 0, None, 0))
            return
        for a in (block, alt):
            if a.dtype != np.uint8 or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]:
                raise ValueError("set_result_memory takes writable, contiguous uint8 arrays")
        _check(self.lib, self.lib.snf_batch_set_result_memory(self._h, C.c_void_p(block.ctypes.data), int(block.nbytes),
                                                              C.c_void_p(alt.ctypes.data), int(alt.nbytes)))
        if not hasattr(self, "_result_mems"):
            self._result_mems = {}
        self._result_mems[(block.ctypes.data, alt.ctypes.data)] = (block, alt)   # (page-locked by the library until the batch is closed)

    def export_device(self, dst_ptr: int, cap_bytes: int) -> dict:
        """Device-to-device copy of the finalized result block [records | read names | ALT bytes] into caller-owned HBM (for
        the RCCL gather); returns its layout (n_calls, rnames_len, alt_pool_len, off_rnames, off_alt, bytes)."""
        lay = abi.snf_export_layout_t()
        _check(self.lib, self.lib.snf_batch_export_device(self._h, C.c_void_p(dst_ptr), int(cap_bytes), C.byref(lay)))
        return {f: int(getattr(lay, f)) for f, _ in abi.snf_export_layout_t._fields_}

    def block_coverage(self, task_index: int, binsize: int, first_bin: int, n_bins: int) -> np.ndarray:
        """Rounded mean depth of `n_bins` coverage bins of `binsize` bp (SNFile.annotate_block_coverages); -1 = beyond
        the padded coverage vector.  Needs call_candidates first."""
        out = np.empty(max(n_bins, 1), np.int32)
        _check(self.lib, self.lib.snf_batch_block_coverage(self._h, task_index, binsize, first_bin, n_bins,
                                                           out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n_bins]

    def coverage_calls(self, task_index: int, svtype, pos, svlen, bnd_is_first, cov):
        """postprocessing.coverage for arbitrary calls of a task: returns (cov [n, 5] updated, status, coverage mean)."""
        i32p, n = C.POINTER(C.c_int32), len(pos)
        a = [np.ascontiguousarray(np.asarray(x, np.int32).reshape(-1)) for x in (svtype, pos, svlen)]
        f = np.ascontiguousarray(np.asarray(bnd_is_first, np.uint8).reshape(-1))
        cv = np.ascontiguousarray(np.asarray(cov, np.int32).reshape(-1)).copy()
        if n == 0:
            a, f, cv = [np.zeros(1, np.int32)] * 3, np.zeros(1, np.uint8), np.zeros(5, np.int32)
        st, mean = C.c_int32(), C.c_double()
        _check(self.lib, self.lib.snf_batch_coverage_calls(self._h, task_index, n, a[0].ctypes.data_as(i32p), a[1].ctypes.data_as(i32p),
                                                           a[2].ctypes.data_as(i32p), f.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                           cv.ctypes.data_as(i32p), C.byref(st), C.byref(mean)))
        return cv[:5 * n].reshape(n, 5), int(st.value), float(mean.value)

    def timings(self) -> list:
        out = []
        for i in range(self.lib.snf_batch_timing_count(self._h)):
            name, ms, nb = C.c_char_p(), C.c_float(), C.c_int64()
            _check(self.lib, self.lib.snf_batch_timing_get(self._h, i, C.byref(name), C.byref(ms), C.byref(nb)))
            out.append((name.value.decode(), float(ms.value), int(nb.value)))
        return out

    def timing_every(self, n: int) -> None:
        """HIP-event brackets around the launches on every n-th pass of this handle (default 8; 1: every pass; 0: never)."""
        self.lib.snf_batch_timing_every(self._h, int(n))

    def timings_mean_reset(self) -> None:
        self.lib.snf_batch_timing_mean_reset(self._h)

    def timings_mean(self) -> list:
        """(name, mean ms per pass, algorithmic bytes) over the passes since `timings_mean_reset`."""
        out = []
        for i in range(self.lib.snf_batch_timing_mean_count(self._h)):
            name, ms, nb, k = C.c_char_p(), C.c_float(), C.c_int64(), C.c_int()
            _check(self.lib, self.lib.snf_batch_timing_mean_get(self._h, i, C.byref(name), C.byref(ms), C.byref(nb), C.byref(k)))
            out.append((name.value.decode(), float(ms.value), int(nb.value)))
        return out


class PendingBatch:
    """A batch that is being opened on a helper thread (`Batch.open_in_background`)."""
    _batch = None
    _err = None
    _thread = None

    def result(self) -> Batch:
        self._thread.join()
        if self._err is not None:
            raise self._err
        b, self._batch = self._batch, None        # (handed over: the caller closes it)
        return b

    def __del__(self):
        # a pending batch nobody asked for any more: its handle goes with it (not at interpreter exit only)
        try:
            if self._thread is not None and self._batch is not None:
                self.discard()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def discard(self):
        self._thread.join()
        if self._batch is not None:
            self._batch.close()
            self._batch = None


def trim_caches(device: int = -1) -> int:
    """Release what the library keeps of finished batches (HBM slabs, pinned buffers, idle streams); bytes released."""
    return int(load().snf_trim_caches(device))


def device_count() -> int:
    return int(load().snf_device_count())


def edit_distance_batch(pairs, device: int = 0, max_dist=None) -> np.ndarray:
    """Global unit-cost edit distance for a list of (bytes, bytes) pairs (edlib.align(a,b)['editDistance']).
    `max_dist` (one int per pair, or one int for all; edlib's `k`): distances beyond it come back as -1 and the alignment
    is banded accordingly; None / negative: exact."""
    lib = load()
    n = len(pairs)
    a_off = np.zeros(n + 1, np.int64)
    b_off = np.zeros(n + 1, np.int64)
    for i, (a, b) in enumerate(pairs):
        a_off[i + 1] = a_off[i] + len(a)
        b_off[i + 1] = b_off[i] + len(b)
    a_pool = np.frombuffer(b"".join(p[0] for p in pairs) or b"\0", np.uint8)
    b_pool = np.frombuffer(b"".join(p[1] for p in pairs) or b"\0", np.uint8)
    out = np.zeros(max(n, 1), np.int32)
    u8p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
    if max_dist is None:
        _check(lib, lib.snf_edit_distance_batch(device, a_pool.ctypes.data_as(u8p), a_off.ctypes.data_as(i64p),
                                                b_pool.ctypes.data_as(u8p), b_off.ctypes.data_as(i64p), n,
                                                out.ctypes.data_as(C.POINTER(C.c_int32))))
    else:
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(max_dist, np.int32), (max(n, 1),)))
        rc = lib.snf_edit_distance_batch_k(device, a_pool.ctypes.data_as(u8p), a_off.ctypes.data_as(i64p),
                                           b_pool.ctypes.data_as(u8p), b_off.ctypes.data_as(i64p), n,
                                           k.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise SnifflesAmdError("snf_edit_distance_batch_k failed (no HIP device?)")
    return out[:n]


def combine_resolve_batch(cfg, problems, device: int = 0) -> None:
    """Run a list of packed resolve_block_groups problems (abi.combine_problem structs); fills their out_group arrays."""
    lib = load()
    if not len(problems):
        return
    arr = problems if isinstance(problems, C.Array) else (abi.snf_combine_problem_t * len(problems))(*problems)
    cs = abi.config_struct(cfg)
    rc = lib.snf_combine_resolve_batch(C.byref(cs), device, arr, len(problems))
    if rc != 0:
        raise SnifflesAmdError("snf_combine_resolve_batch failed (no HIP device, or invalid sample ids)")


def combine_call_groups(cfg, group_off, member, cand, cand_win, group_win_hi, win_bin, win_thr, device: int = 0):
    """`SVGroup.call` and the keep / flush walk for all groups of a merge (snf_combine_call_groups).  cand: abi.GROUP_CAND_DTYPE
    records; member / group_off: the groups' candidates in add order.  Returns (out: abi.GROUP_OUT_DTYPE per group,
    chosen: uint8 per member, pos_mean: float64 per member)."""
    lib = load()
    group_off = np.ascontiguousarray(group_off, np.int64)
    n_groups = len(group_off) - 1
    member = np.ascontiguousarray(member, np.int32)
    cand = np.ascontiguousarray(cand, abi.GROUP_CAND_DTYPE)
    cand_win = np.ascontiguousarray(cand_win, np.int32)
    group_win_hi = np.ascontiguousarray(group_win_hi, np.int32)
    win_bin = np.ascontiguousarray(win_bin, np.int32)
    win_thr = np.ascontiguousarray(win_thr, np.float64)
    out = np.zeros(max(n_groups, 0), abi.GROUP_OUT_DTYPE)
    chosen = np.zeros(len(member), np.uint8)
    pos_mean = np.zeros(len(member), np.float64)
    if n_groups <= 0:
        return out, chosen, pos_mean
    gc = abi.group_call_config(cfg)
    p = lambda a: a.ctypes.data  # noqa: E731
    rc = lib.snf_combine_call_groups(C.byref(gc), device, n_groups, p(group_off), p(member), len(cand), p(cand), p(cand_win), p(group_win_hi),
                                     len(win_bin), p(win_bin), p(win_thr), p(out), p(chosen), p(pos_mean))
    if rc != 0:
        raise SnifflesAmdError("snf_combine_call_groups failed (no HIP device, invalid membership, or a candidate behind its group's flush)")
    return out, chosen, pos_mean


def combine_last_stats(device: int = 0) -> dict:
    """Kernel time and alignment counters of the last combine_resolve_batch on `device`."""
    lib = load()
    ms, st = C.c_double(), (C.c_int64 * 4)()
    if lib.snf_combine_last_stats(device, C.byref(ms), st) != 0:
        raise SnifflesAmdError("snf_combine_last_stats failed")
    return dict(kernel_ms=float(ms.value), alignments=int(st[0]), aligned_bytes=int(st[1]), dp_cells=int(st[2]), staged_bytes=int(st[3]))


def genotype_batch(cfg, records: np.ndarray, device: int = 0) -> np.ndarray:
    """genotype_sv (genotyping.py:62-241) over a structured array of call records (abi.CALL_DTYPE), in place."""
    lib = load()
    records = np.ascontiguousarray(records, abi.CALL_DTYPE)
    cs = abi.config_struct(cfg)
    _check(lib, lib.snf_genotype_batch(C.byref(cs), device, records.ctypes.data_as(C.POINTER(abi.snf_call_t)), len(records)))
    return records
