"""One GPU server process per device for many worker processes ("a spawn-based GPU server process per device").

The reference's deployment is a pool of worker PROCESSES that pull contig tasks (`sniffles:495-530`, `parallel.py:585-769`), at most
one per contig: two dozen.  Each process that opens the device itself brings a HIP context and its own hardware queues; beyond eight
or so the driver time-slices them and a pass that takes a millisecond alone takes tens (24 workers: 0.44-0.9 s for a genome that 8
workers finish in 0.08 s; bounding the number of passes in flight does not help - it is the contexts, not the work).  Here ONE process
owns the device: workers hand their task inputs over in shared memory, the server runs whatever has arrived as ONE device batch
(`lib.Batch` - the library's native shape: all tasks of a batch share the launches) and lands the finalized result block in a
shared-memory segment every worker of the batch maps; a worker turns ITS records into objects, in parallel with the others.

    srv = server.start(device=0)                      # parent: spawns the server, returns when it listens (srv.address)
    os.environ["SNF_GPU_SERVER"] = srv.address        # workers (any start method): sniffles_amd.parallel.Task goes through it
    ...
    srv.stop()

Protocol (multiprocessing.connection over a Unix socket; messages are small dicts, bulk data lives in /dev/shm):
  worker -> server  {"op": "task", "seg": name, "fields": [(name, dtype, offset, count)], "meta": {...}, "cfg": bytes of snf_config_t}
  server -> worker  {"seg": name, "lo": .., "hi": .., "n_calls": .., "rnames_len": .., "alt_len": .., "off_rnames": .., "off_alt": ..,
                     "status": .., "coverage_average_total": ..}         (or {"error": text})
  worker -> server  {"op": "release", "seg": name}    the worker no longer reads the result segment (refcounted, reused)
What a worker gets is the FINALIZED record table (`Task.finalize_candidates`' state): `Task.call_candidates` hands out stand-ins
(sv.LazySource) over it, so a candidate touched between the two calls already carries its final fields - the one difference from the
in-process path.  `cluster.resolve` (seam B3) and the SNF writer's coverage bins need the batch itself and are not served through a
server.
"""
from __future__ import annotations

import os
import threading

import numpy as np

from . import abi
from .soa import LEAD_FIELDS, TaskInput

_READ_FIELDS = [("read_start", np.int32), ("read_end", np.int32), ("read_hp", np.uint8)]
_OPT_FIELDS = [("tr_start", np.int32), ("tr_end", np.int32), ("nmask_start", np.int32), ("nmask_end", np.int32)]


# ---------------------------------------------------------------------------------------------- shared-memory segments
class Segment:
    """A file under /dev/shm mapped as a uint8 array (created or attached)."""

    def __init__(self, name: str, size: int = 0, create: bool = False, directory: str = "/dev/shm"):
        self.name, self.path = name, os.path.join(directory, name)
        if create:
            with open(self.path, "wb") as f:
                f.truncate(max(int(size), 4096))
        self.size = os.path.getsize(self.path)
        self.buf = np.memmap(self.path, np.uint8, "r+", shape=(self.size,))

    def unlink(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass


def _pack_task(ti: TaskInput, seg_holder: list, tag: str):
    """The arrays of a task input back to back (256-byte aligned) in the worker's input segment (grown when needed)."""
    ti.check_layout()
    arrays = [(n, np.ascontiguousarray(ti.leads[n], dt)) for n, dt in LEAD_FIELDS]
    arrays.append(("seq_pool", np.ascontiguousarray(ti.seq_pool, np.uint8)))
    arrays += [(n, np.ascontiguousarray(getattr(ti, n), dt)) for n, dt in _READ_FIELDS]
    for n, dt in _OPT_FIELDS:
        a = getattr(ti, n, None)
        if a is not None:
            arrays.append((n, np.ascontiguousarray(a, dt)))
    total, fields = 0, []
    for n, a in arrays:
        fields.append((n, a.dtype.str, total, int(a.shape[0])))
        total += (a.nbytes + 255) & ~255
    if not seg_holder or seg_holder[0].size < total:
        if seg_holder:
            seg_holder[0].unlink()
            seg_holder.clear()
        seg_holder.append(Segment(f"{tag}_in_{os.urandom(3).hex()}", total * 5 // 4 + 4096, create=True))
    buf = seg_holder[0].buf
    for (n, a), (_, _, off, _) in zip(arrays, fields):
        if a.nbytes:
            buf[off:off + a.nbytes] = a.view(np.uint8).reshape(-1)
    null_rank = ti.ps_names.index("NULL") if ti.ps_names is not None and "NULL" in ti.ps_names else -1
    meta = dict(task_id=int(ti.task_id), contig=ti.contig, contig_len=int(ti.contig_len), sv_id_start=int(ti.sv_id_start),
                qc_nm_threshold=float(ti.qc_nm_threshold), ps_null_rank=null_rank)
    return seg_holder[0].name, fields, meta


class _ServedTaskInput(TaskInput):
    """The server's view of a worker's task: columns are views of the worker's segment; the name tables stay with the worker (the
    library only needs the rank of "NULL" among the phase sets)."""
    ps_null_rank_value = -1

    def check_layout(self) -> None:          # built from typed views of known sizes
        pass


def _unpack_task(buf, fields, meta) -> TaskInput:
    cols = {}
    for n, dt, off, cnt in fields:
        dt = np.dtype(dt)
        cols[n] = buf[off:off + cnt * dt.itemsize].view(dt)
    ti = _ServedTaskInput(task_id=meta["task_id"], contig=meta["contig"], contig_len=meta["contig_len"], sv_id_start=meta["sv_id_start"],
                          leads={n: cols[n] for n, _ in LEAD_FIELDS}, seq_pool=cols["seq_pool"],
                          read_start=cols["read_start"], read_end=cols["read_end"], read_hp=cols["read_hp"],
                          tr_start=cols.get("tr_start"), tr_end=cols.get("tr_end"), qc_nm_threshold=meta["qc_nm_threshold"])
    ti.nmask_start, ti.nmask_end = cols.get("nmask_start"), cols.get("nmask_end")
    # abi.task_struct asks the phase-set names for the rank of "NULL": a one-entry stand-in at that rank
    r = meta["ps_null_rank"]
    ti.ps_names = None if r < 0 else _NullAt(r)
    return ti


class _NullAt:
    def __init__(self, rank):
        self.rank = rank

    def __contains__(self, x):
        return x == "NULL"

    def index(self, x):
        if x != "NULL":
            raise ValueError(x)
        return self.rank


# ---------------------------------------------------------------------------------------------- server process
def _serve(address: str, device: int, init: str, extra_path, ready, arena_mb: int = 0):
    import queue
    import sys
    if arena_mb > 0:       # (read by the library at its first upload)
        os.environ.setdefault("SNF_STAGE_ARENA_MB", str(int(arena_mb)))
    from multiprocessing.connection import Listener
    for p in extra_path or ():
        if p not in sys.path:
            sys.path.insert(0, p)
    if init:                                   # e.g. "emu.emu:lib": the host tier of the test suite becomes the library
        mod, _, fn = init.partition(":")
        getattr(__import__(mod, fromlist=[fn]), fn)()
    from . import lib
    lib.load()
    q = queue.Queue()
    listener = Listener(address, family="AF_UNIX")
    tag = "snfsrv_%d_%s" % (os.getpid(), os.urandom(3).hex())
    in_maps = {}                               # worker input segments, by name
    free, busy = [], {}                        # result segments: free list; name -> [segment, readers left]
    lock = threading.Lock()
    stop = threading.Event()

    held = {}                                  # id(conn) -> {result segment: replies not yet released}: a worker that dies gives them back
    conn_in = {}                               # id(conn) -> name of the worker's current input segment

    def give_back(name, times=1):
        with lock:
            ent = busy.get(name)
            if ent is not None:
                ent[1] -= times
                if ent[1] <= 0:
                    del busy[name]
                    free.append(ent[0])

    def reader(conn):
        mine = held.setdefault(id(conn), {})
        try:
            while True:
                m = conn.recv()
                if m.get("op") == "release":
                    with lock:
                        if mine.get(m["seg"], 0) > 0:
                            mine[m["seg"]] -= 1
                    give_back(m["seg"])
                elif m.get("op") == "stop":
                    stop.set(); q.put(None)
                else:
                    with lock:                 # a worker that grew its input segment has unlinked the old one: let go of the mapping
                        old = conn_in.get(id(conn))
                        if old is not None and old != m["seg"]:
                            in_maps.pop(old, None)
                        conn_in[id(conn)] = m["seg"]
                    q.put((conn, m))
        except (EOFError, OSError):
            pass
        finally:                               # the worker is gone: what it still held, and the mapping of its input segment
            with lock:
                left = {k: v for k, v in mine.items() if v > 0}
                mine.clear()
                held.pop(id(conn), None)
                old = conn_in.pop(id(conn), None)
                if old is not None:
                    in_maps.pop(old, None)
            for name, times in left.items():
                give_back(name, times)

    def acceptor():
        while not stop.is_set():
            try:
                conn = listener.accept()
            except OSError:
                break
            threading.Thread(target=reader, args=(conn,), daemon=True).start()
    threading.Thread(target=acceptor, daemon=True).start()
    ready.set()

    def take_segment(size):
        with lock:
            for k, s in enumerate(free):
                if s.size >= size:
                    return free.pop(k)
        return Segment(f"{tag}_out{len(busy) + len(free)}_{os.urandom(2).hex()}", size * 5 // 4 + (1 << 20), create=True)

    sends = {}                                  # one lock per connection: replies of several batches must not interleave

    def send(conn, msg):
        lk = sends.setdefault(id(conn), threading.Lock())
        with lk:
            try:
                conn.send(msg)
            except (OSError, ValueError):
                pass

    prof = os.environ.get("SNF_PROF") is not None

    def run_group(group):
        import time
        t0 = time.perf_counter()
        try:
            tis = []
            for conn, m in group:
                with lock:
                    seg = in_maps.get(m["seg"])
                    if seg is None or seg.size < max(off + cnt * np.dtype(dt).itemsize for _, dt, off, cnt in m["fields"]):
                        seg = in_maps[m["seg"]] = Segment(m["seg"])
                tis.append(_unpack_task(seg.buf, m["fields"], m["meta"]))
            cs = abi.snf_config_t.from_buffer_copy(group[0][1]["cfg"])
            t1 = time.perf_counter()
            with lib.Batch(cs, tis, device=device) as b:
                t2 = time.perf_counter()
                b.set_output(abi.OUT_CANDIDATES)
                b.run_pass()
                res = b.fetch(1, copy=False)
                t3 = time.perf_counter()
                n = len(res.calls)
                rec_bytes = n * abi.CALL_DTYPE.itemsize
                off_rn = (rec_bytes + 255) & ~255
                off_alt = (off_rn + 4 * len(res.rnames) + 255) & ~255
                total = off_alt + len(res.alt_pool) + 256
                out = take_segment(total)
                if rec_bytes:
                    out.buf[:rec_bytes] = res.calls.view(np.uint8).reshape(-1)
                if len(res.rnames):
                    out.buf[off_rn:off_rn + 4 * len(res.rnames)] = res.rnames.view(np.uint8).reshape(-1)
                if len(res.alt_pool):
                    out.buf[off_alt:off_alt + len(res.alt_pool)] = res.alt_pool
                with lock:
                    busy[out.name] = [out, len(group)]
                    for conn, m in group:
                        h = held.get(id(conn))
                        if h is None:          # (the worker left while its batch ran)
                            busy[out.name][1] -= 1
                        else:
                            h[out.name] = h.get(out.name, 0) + 1
                    if busy[out.name][1] <= 0:
                        del busy[out.name]
                        free.append(out)
                for t, (conn, m) in enumerate(group):
                    send(conn, dict(seg=out.name, lo=int(res.task_call_off[t]), hi=int(res.task_call_off[t + 1]), n_calls=n,
                                    rnames_len=int(len(res.rnames)), alt_len=int(len(res.alt_pool)), off_rnames=off_rn, off_alt=off_alt,
                                    status=int(res.task_status[t]), coverage_average_total=float(res.coverage_average_total[t]),
                                    batch_tasks=len(group)))
                if prof:
                    print(f"[SNF_PROF] server batch: {len(group)} tasks, {sum(t.n_leads for t in tis)} leads, {n} calls | map {1e3 * (t1 - t0):.2f} "
                          f"upload {1e3 * (t2 - t1):.2f} pass {1e3 * (t3 - t2):.2f} copy + replies {1e3 * (time.perf_counter() - t3):.2f} ms",
                          file=sys.stderr, flush=True)
        except BaseException as e:  # noqa: BLE001 - reported to the workers of the batch, the server goes on
            import traceback
            text = f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"
            for conn, m in group:
                send(conn, dict(error=text))

    gather_lock = threading.Lock()        # one dispatcher gathers at a time; the other one is running its batch on the device

    def dispatcher():
        import time
        while not stop.is_set():
            with gather_lock:
                item = q.get()
                if item is None:
                    q.put(None)            # (the other dispatcher sees it too)
                    return
                group, other = [item], []
                # everything that arrives for the same configuration within a moment runs as ONE device batch: workers released by a
                # common event submit within a fraction of a millisecond of each other, and a batch of one costs what a batch of
                # twenty does (the launches are shared); the wait ends as soon as nothing new has come for 0.2 ms, after 2 ms at most
                t_end, quiet = time.perf_counter() + 2e-3, 0
                while len(group) < 256 and quiet < 2 and time.perf_counter() < t_end:
                    try:
                        nxt = q.get(timeout=2e-4)
                    except queue.Empty:
                        quiet += 1
                        continue
                    quiet = 0
                    if nxt is None:
                        q.put(None); stop.set(); break
                    (group if nxt[1]["cfg"] == item[1]["cfg"] else other).append(nxt)
                for x in other:
                    q.put(x)
            run_group(group)

    workers = [threading.Thread(target=dispatcher, daemon=True) for _ in range(2)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    try:
        listener.close()
    except OSError:
        pass
    for s in list(free) + [e[0] for e in busy.values()]:
        s.unlink()
    try:
        os.unlink(address)
    except OSError:
        pass


class ServerHandle:
    def __init__(self, process, address):
        self.process, self.address = process, address

    def stop(self, timeout: float = 10.0):
        try:
            from multiprocessing.connection import Client as _C
            c = _C(self.address, family="AF_UNIX")
            c.send(dict(op="stop"))
            c.close()
        except OSError:
            pass
        self.process.join(timeout)
        if self.process.is_alive():
            self.process.terminate()


def start(device: int = 0, address: str = None, init: str = None, extra_path=None, timeout: float = 120.0, arena_mb: int = 768) -> ServerHandle:
    """Spawn the server of `device` (a fresh interpreter: it opens the device itself) and wait until it listens.  `arena_mb`: pinned
    staging memory the server reserves at its first upload (SNF_STAGE_ARENA_MB; a 30x human genome in one batch stages 0.46 GB) - a batch
    beyond it makes the arena grow when it arrives, which stalls the whole server for the ~0.15 s the pinning takes."""
    import multiprocessing as mp
    import tempfile
    address = address or os.path.join(tempfile.gettempdir(), "snf_gpu_%d_%s.sock" % (os.getpid(), os.urandom(3).hex()))
    ctx = mp.get_context("spawn")
    ready = ctx.Event()
    p = ctx.Process(target=_serve, args=(address, device, init, list(extra_path or ()), ready, int(arena_mb)), daemon=True)
    p.start()
    if not ready.wait(timeout):
        p.terminate()
        raise RuntimeError("the GPU server did not come up")
    return ServerHandle(p, address)


# ---------------------------------------------------------------------------------------------- worker side
class Reply:
    """The finalized result of one task as a worker sees it: `result` (an abi.Result-shaped view: the task's own records, the
    batch's pools), status, coverage average.  `release()` when nothing reads the views any more."""

    def __init__(self, client, msg):
        self.client, self.msg = client, msg
        seg = client._map(msg["seg"], msg["off_alt"] + msg["alt_len"])
        n, lo, hi = msg["n_calls"], msg["lo"], msg["hi"]
        calls = seg.buf[:n * abi.CALL_DTYPE.itemsize].view(abi.CALL_DTYPE)
        r = object.__new__(abi.Result)
        r.calls = calls[lo:hi]
        r.rnames = seg.buf[msg["off_rnames"]:msg["off_rnames"] + 4 * msg["rnames_len"]].view(np.uint32)
        r.alt_pool = seg.buf[msg["off_alt"]:msg["off_alt"] + msg["alt_len"]]
        r.task_status = np.asarray([msg["status"]], np.int32)
        r.task_call_off = np.asarray([0, hi - lo], np.int64)
        r.coverage_average_total = np.asarray([msg["coverage_average_total"]], np.float64)
        self.result = r
        self._released = False

    def release(self):
        if not self._released:
            self._released = True
            self.result = None
            self.client._send(dict(op="release", seg=self.msg["seg"]))


class Client:
    """A worker's connection to the server of its device (one per process and address; thread-safe for one request at a time)."""

    def __init__(self, address: str):
        from multiprocessing.connection import Client as _C
        self.address = address
        self.conn = _C(address, family="AF_UNIX")
        self.lock = threading.Lock()
        self._in = []
        self._maps = {}
        self._tag = "snfwrk_%d_%s" % (os.getpid(), os.urandom(3).hex())

    def _send(self, msg):
        with self.lock:
            self.conn.send(msg)

    def _map(self, name, need):
        s = self._maps.get(name)
        if s is None or s.size < need:
            s = self._maps[name] = Segment(name)
        return s

    def run_task(self, cfg, ti: TaskInput) -> Reply:
        cs = cfg if isinstance(cfg, abi.snf_config_t) else abi.config_struct(cfg)
        import time
        with self.lock:
            t0 = time.perf_counter()
            name, fields, meta = _pack_task(ti, self._in, self._tag)
            self.last_pack_ms = (time.perf_counter() - t0) * 1e3
            self.conn.send(dict(op="task", seg=name, fields=fields, meta=meta, cfg=bytes(cs)))
            msg = self.conn.recv()
        if "error" in msg:
            raise RuntimeError("GPU server: " + msg["error"])
        return Reply(self, msg)

    def close(self):
        try:
            self.conn.close()
        except OSError:
            pass
        for s in self._in:
            s.unlink()
        self._in = []


_clients = {}


def client(address: str = None):
    """The process-wide client of `address` (default: $SNF_GPU_SERVER); None when no server is configured."""
    address = address or os.environ.get("SNF_GPU_SERVER")
    if not address:
        return None
    key = (os.getpid(), address)
    c = _clients.get(key)
    if c is None:
        c = _clients[key] = Client(address)
        import atexit
        atexit.register(c.close)
    return c
