"""Output record types with the reference's names and fields (reference `src/sniffles/sv.py:31-223`).

`SVCall` / `SVCallBNDInfo` here are plain dataclasses carrying the same attributes the reference's
downstream code reads (VCF writer, SNF writer, CallTask.execute).  When the library is used inside the
reference package, pass the reference's own classes to `materialize(..., svcall_cls=sniffles.sv.SVCall,
bnd_cls=sniffles.sv.SVCallBNDInfo)` so pickles carry the `sniffles.sv` module path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

from .abi import FILTERS, Result, none_if_nan
from .records import bnd_alt, call_id
from .soa import SVTYPES

TYPES = ["INS", "DEL", "DUP", "INV", "BND"]
SINGLE_TYPES = ["SINGLE_LEFT", "SINGLE_RIGHT"]
ALL_TYPES = TYPES + SINGLE_TYPES


@dataclass
class SVCallBNDInfo:
    mate_contig: str
    mate_ref_start: int
    is_first: bool
    is_reverse: bool


@dataclass
class SVCallPostprocessingInfo:
    """Stands in for `SVCallPostprocessingInfo(cluster=...)`: the cluster lives in HBM; this is its handle."""
    batch: object
    index: int

    @property
    def cluster(self):
        raise AttributeError("cluster leads stay on the GPU; use the call's fields (support, rnames, ...) instead")


class ForwardDifferenceWelford:
    """State of the reference's relative forward-difference sampler (sv.py:51-87).  Nothing on the path ever feeds it
    (`qc_coverage_samples` therefore always reports `(True, None)`, postprocessing.py:373); it exists so that pickled
    calls carry the attribute the reference's `SVCall` has (SNF blocks, sniffles_amd/snf.py)."""

    def __init__(self):
        self.n = 0
        self.m1 = 0
        self.m2 = 0
        self.last = None

    def push(self, value):
        if self.last is None:
            self.last = value
            return
        rel = (value - self.last) / (self.last + 1e-10)
        k = self.n
        self.n = k + 1
        d = rel - self.m1
        step = d / self.n
        self.m1 += step
        self.m2 += d * step * k
        self.last = value

    @property
    def mean(self):
        return self.m1 if self.n else None

    @property
    def variance(self):
        return self.m2 / self.n if self.n >= 2 else None


@dataclass
class SVCall:
    contig: str
    pos: int
    id: str
    ref: str
    alt: str
    qual: int
    filter: str
    info: dict
    svtype: str
    svlen: int
    end: int
    genotypes: dict
    precise: bool
    support: int
    rnames: Optional[list]
    qc: bool
    nm: float
    postprocess: Optional[SVCallPostprocessingInfo]
    svlens: Optional[list] = None
    fwd: int = None
    rev: int = None
    forward_difference_sampler: ForwardDifferenceWelford = field(default_factory=ForwardDifferenceWelford)
    coverage_upstream: int = 0
    coverage_downstream: int = 0
    coverage_start: int = 0
    coverage_center: int = 0
    coverage_end: int = 0
    sample_internal_id: int = None
    bnd_info: SVCallBNDInfo = None
    support_inline: int = None
    support_splits: int = None
    raw_vcf_line: Optional[str] = None
    raw_vcf_line_index: Optional[int] = None

    def set_info(self, k, v):
        self.info[k] = v

    def get_info(self, k):
        return self.info[k] if k in self.info else None

    def has_info(self, k):
        return k in self.info

    @property
    def is_single_break(self) -> bool:
        return self.svtype.startswith("SINGLE")

    def finalize(self):
        self.postprocess = None

    def qc_coverage_samples(self):
        var = self.forward_difference_sampler.variance
        return (True, None) if var is None else (var < 0.3, float(var))


_QC_SV_EARLY_EXIT = frozenset(("STDEV_POS", "STDEV_LEN", "SINGLE_BREAK", "SVLEN_MIN", "STRAND_BND", "COV_CHANGE_DEL",
                               "COV_CHANGE_DUP", "COV_CHANGE_INS", "INLINE_SA"))


def _ps(code, ti):
    if code == -1:
        return None
    if code == -2:
        return "NULL"
    return ti.ps_name(int(code))


def fill_candidate(call, res: Result, i: int, ti, bnd_cls=SVCallBNDInfo):
    """Candidate-stage fields of SVCall `call` from record i (sv.call_from, sv.py:497-598)."""
    c = res.calls[i]
    svtype = SVTYPES[int(c["svtype"])]
    call.contig, call.pos, call.end, call.svtype, call.svlen = ti.contig, int(c["pos"]), int(c["end"]), svtype, int(c["svlen"])
    call.id = call_id(svtype, int(c["sv_id"]), ti.task_id)
    call.ref, call.alt = "N", f"<{svtype}>"
    call.qual, call.filter, call.qc = int(c["qual"]), FILTERS[int(c["filter"])], bool(c["qc"])
    call.precise, call.support = bool(c["precise"]), int(c["support"])
    call.fwd, call.rev, call.nm = int(c["fwd"]), int(c["rev"]), float(c["nm"])
    call.rnames = [ti.qname(int(q)) for q in res.rn(i)]
    call.genotypes = dict()
    info = dict()
    if svtype == "BND":
        bi = bnd_cls(mate_contig=ti.contig_name(int(c["mate_contig"])), mate_ref_start=int(c["mate_ref_start"]),
                     is_first=bool(c["bnd_is_first"]), is_reverse=bool(c["bnd_is_reverse"]))
        call.bnd_info = bi
        call.alt = bnd_alt(bi.mate_contig, bi.mate_ref_start, bi.is_first, bi.is_reverse)
        info["CHR2"] = bi.mate_contig
    elif svtype == "INS":
        info["SUPPORT_LONG"] = int(c["support_long"])
    elif svtype == "DEL":
        info["SUPPORT_SA"] = int(c["support_sa"])
    # util.stdev returns the int 0 for fewer than two values (util.py:25-27) and a float otherwise; the trimmed list
    # is that short only for a single lead (fwd + rev = len(leads)).  0 == 0.0, but the VCF writer prints them differently.
    single = int(c["fwd"]) + int(c["rev"]) < 2
    info["STDEV_POS"] = 0 if single else float(c["stdev_pos"])
    sl = none_if_nan(c["stdev_len"])
    if sl is not None:
        info["STDEV_LEN"] = 0 if single else sl
    call.info = info
    (call.coverage_upstream, call.coverage_start, call.coverage_center, call.coverage_end,
     call.coverage_downstream) = (int(x) for x in c["cov"])
    return call


def fill_final(call, res: Result, i: int, ti):
    """Fields set by Task.finalize_candidates (parallel.py:129-201) from record i."""
    c = res.calls[i]
    call.qc, call.filter = bool(c["qc"]), FILTERS[int(c["filter"])]
    # qc_sv sets info["COVERAGE_VAR"] = None (qc_coverage_samples() always yields (True, None), postprocessing.py:373-374)
    # unless it left earlier through one of its own filters (postprocessing.py:209-371); no later stage uses those
    # names, so the final filter tells which (residual: a GT_FAILED call hides an earlier exit - the key is then present
    # where the reference may omit it; value None either way, `get_info` and the VCF writer do not distinguish).
    if call.filter not in _QC_SV_EARLY_EXIT:
        call.info["COVERAGE_VAR"] = None
    if c["ph_set"]:
        call.info["PHASE"] = (f"{int(c['ph_hp'])},{_ps(int(c['ph_ps']), ti)},{int(c['ph_hp_support'])},"
                              f"{int(c['ph_ps_support'])},{'PASS' if c['ph_hp_pass'] else 'FAIL'},"
                              f"{'PASS' if c['ph_ps_pass'] else 'FAIL'}")
    if c["gt_set"]:
        hp = None if c["gt_hp"] < 0 else str(int(c["gt_hp"]))
        call.genotypes[0] = (int(c["gt_a"]), int(c["gt_b"]), int(c["gt_gq"]), int(c["gt_dr"]), int(c["gt_dv"]),
                             (hp, _ps(int(c["gt_ps"]), ti)))
        call.info["VAF"] = float(c["vaf"])
    alt = res.alt(i)
    if alt is not None:
        call.alt = alt
    return call


# ---- whole result sets at once: one C-level conversion of the record table to Python scalars instead of ~40 numpy scalar
# reads per call (materialising the objects is the slowest part of a task on the host; the device pass takes milliseconds)
def _columns(res: Result, lo: int, hi: int):
    names = res.calls.dtype.names
    return {n: k for k, n in enumerate(names)}, res.calls[lo:hi].tolist()


_fast = None


def _load_fast():
    """The C materialiser (csrc/snf_pyfast.c), built in-tree by `sniffles_amd.build`; None when it is not there (the
    pure-Python twins below do the same work, ~10x slower)."""
    global _fast
    if _fast is None:
        try:
            from . import _snf_fast as m
            _fast = m
        except ImportError:
            _fast = False
    return _fast or None


class no_gc:
    """Cyclic garbage collection off while a C pass creates objects in bulk: every generation-2 sweep walks all live containers
    (the millions of dicts / lists / tuples of the calls already built) and finds nothing - the new objects are acyclic."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()


def materialize_candidates(res: Result, ti, lo: int, hi: int, svcall_cls=SVCall, bnd_cls=SVCallBNDInfo, post_cls=None, batch=None, idx=None) -> list:
    """`fill_candidate(new_call(), res, i, ti)` for i in [lo, hi), same objects.  With `post_cls` every call also gets its
    `postprocess = post_cls(batch=batch, index=i - lo)` handle.  `idx` (int64, relative to lo): only those records, in that order."""
    import numpy as np
    fast = _load_fast()
    if fast is not None and (ti.qnames is None or isinstance(ti.qnames, list)) and (ti.contig_names is None or isinstance(ti.contig_names, list)):
        with no_gc():
            extra = () if idx is None else (np.ascontiguousarray(idx, np.int64),)
            return fast.materialize(svcall_cls, bnd_cls, ForwardDifferenceWelford, post_cls, batch, np.ascontiguousarray(res.calls), lo, hi,
                                    np.ascontiguousarray(res.rnames, np.uint32), ti.qnames, ti.contig, ti.task_id, ti.contig_names, FILTERS, *extra)
    if idx is not None:
        out = []
        for i in np.asarray(idx).tolist():
            c = materialize_candidates_py(res, ti, lo + int(i), lo + int(i) + 1, svcall_cls, bnd_cls)[0]
            if post_cls is not None:
                c.postprocess = post_cls(batch=batch, index=int(i))
            out.append(c)
        return out
    out = materialize_candidates_py(res, ti, lo, hi, svcall_cls, bnd_cls)
    if post_cls is not None:
        for i, c in enumerate(out):
            c.postprocess = post_cls(batch=batch, index=i)
    return out


def materialize_candidates_py(res: Result, ti, lo: int, hi: int, svcall_cls=SVCall, bnd_cls=SVCallBNDInfo) -> list:
    K, rows = _columns(res, lo, hi)
    k_svtype, k_pos, k_end, k_svlen, k_svid, k_qual, k_filter, k_qc = (K[n] for n in ("svtype", "pos", "end", "svlen", "sv_id", "qual", "filter", "qc"))
    k_prec, k_sup, k_fwd, k_rev, k_nm, k_cov = (K[n] for n in ("precise", "support", "fwd", "rev", "nm", "cov"))
    k_mc, k_mp, k_bf, k_br, k_sl, k_ssa, k_sp, k_sln, k_ro, k_rl = (K[n] for n in ("mate_contig", "mate_ref_start", "bnd_is_first", "bnd_is_reverse", "support_long", "support_sa", "stdev_pos", "stdev_len", "rn_off", "rn_len"))
    rn = res.rnames.tolist()
    qn = ti.qnames
    contig, task_id = ti.contig, ti.task_id
    out = []
    for r in rows:
        svtype = SVTYPES[r[k_svtype]]
        fwd, rev = r[k_fwd], r[k_rev]
        ids = rn[r[k_ro]:r[k_ro] + r[k_rl]]
        info = {}
        alt, bi = f"<{svtype}>", None
        if svtype == "BND":
            bi = bnd_cls(mate_contig=ti.contig_name(r[k_mc]), mate_ref_start=r[k_mp], is_first=bool(r[k_bf]), is_reverse=bool(r[k_br]))
            alt = bnd_alt(bi.mate_contig, bi.mate_ref_start, bi.is_first, bi.is_reverse)
            info["CHR2"] = bi.mate_contig
        elif svtype == "INS":
            info["SUPPORT_LONG"] = r[k_sl]
        elif svtype == "DEL":
            info["SUPPORT_SA"] = r[k_ssa]
        single = fwd + rev < 2
        info["STDEV_POS"] = 0 if single else r[k_sp]
        sl = r[k_sln]
        if sl == sl:            # not NaN
            info["STDEV_LEN"] = 0 if single else sl
        cov = r[k_cov]
        if not isinstance(cov, (list, tuple)):      # a sub-array field comes back as an ndarray: Python ints like everything else
            cov = cov.tolist()
        call = svcall_cls(contig=contig, pos=r[k_pos], id=f"{svtype}.{r[k_svid]:X}S{task_id:X}", ref="N", alt=alt, qual=r[k_qual],
                          filter=FILTERS[r[k_filter]], info=info, svtype=svtype, svlen=r[k_svlen], end=r[k_end], genotypes={},
                          precise=bool(r[k_prec]), support=r[k_sup],
                          rnames=[qn[q] for q in ids] if qn is not None else [f"q{q}" for q in ids],
                          qc=bool(r[k_qc]), nm=r[k_nm], postprocess=None)
        call.fwd, call.rev, call.bnd_info = fwd, rev, bi
        (call.coverage_upstream, call.coverage_start, call.coverage_center, call.coverage_end, call.coverage_downstream) = cov
        out.append(call)
    return out


def apply_final(calls: list, res: Result, ti, lo: int = 0, finalize: bool = False, idx=None) -> None:
    """`fill_final(call, res, lo + k, ti)` for every call of the list (`idx`: call k takes record lo + idx[k]).  `finalize`: followed
    by `call.finalize()` (the reference's `Task.finalize_candidates` ends with it, parallel.py:199-200) - in the same pass over the
    calls when their classes keep `SVCall.finalize` (postprocess = None), by calling the method otherwise."""
    import numpy as np
    fast = _load_fast()
    plain = finalize and all(getattr(t, "finalize", None) is SVCall.finalize for t in {type(c) for c in calls})
    if fast is not None and (ti.ps_names is None or isinstance(ti.ps_names, list)):
        with no_gc():
            if idx is None:
                fast.apply_final(calls, np.ascontiguousarray(res.calls), lo, np.ascontiguousarray(res.alt_pool, np.uint8), ti.ps_names, FILTERS,
                                 _QC_SV_EARLY_EXIT, plain)
            else:
                fast.apply_final(calls, np.ascontiguousarray(res.calls), lo, np.ascontiguousarray(res.alt_pool, np.uint8), ti.ps_names, FILTERS,
                                 _QC_SV_EARLY_EXIT, plain, np.ascontiguousarray(idx, np.int64))
    else:
        if idx is None:
            apply_final_py(calls, res, ti, lo)
        else:
            for c, i in zip(calls, np.asarray(idx).tolist()):
                fill_final(c, res, lo + int(i), ti)
        plain = False
    if finalize and not plain:
        for c in calls:
            c.finalize()


# ---- lazy calls: `Task.call_candidates` hands out a real list whose elements BECOME `SVCall` objects when they are first touched.
# The reference's consumers (CallTask.execute, parallel.py:265-271) read `.qc` of every candidate and everything else only of the
# calls they keep; a 30x genome has 94 k candidates of which 26.8 k are kept, and creating + finalizing + freeing an object with 33
# attributes costs ~2.3 us.  A stand-in is an instance of a subclass of the call class whose dict holds only `qc`, its source and its
# place there; any other access - attribute, method, assignment, pickling, `__dict__` - first turns it into the real thing: the full
# instance dict (the very dict `materialize_candidates` + `apply_final` build) replaces the stand-in's and its class is set to the
# call class, after which nothing distinguishes it from an eagerly built call.  The source turns stand-ins into calls in bulk: the
# first touch materialises every stand-in a consumer is going to touch (those with `qc` set, or all under `no_qc`) in one C pass.
_LAZY_CLASSES = {}


def _raw_dict(obj) -> dict:
    """The instance dict itself (a stand-in's class answers `__dict__` with the filled object's)."""
    get = getattr(type(obj), "_lz_rawdict", None)
    return get(obj) if get is not None else object.__getattribute__(obj, "__dict__")


def _lazy_fill(obj) -> None:
    src = _raw_dict(obj).get("_lz")
    if src is not None:
        src.fill(obj)


def lazy_class(cls):
    """The stand-in class of call class `cls` (cached)."""
    lz = _LAZY_CLASSES.get(cls)
    if lz is not None:
        return lz
    import dataclasses

    def __getattr__(self, name):          # only reached when the normal look-up fails: a field that has no class-level default
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if _raw_dict(self).get("_lz") is None:
            raise AttributeError(name)
        _lazy_fill(self)
        return getattr(self, name)

    def __setattr__(self, name, value):
        _lazy_fill(self)
        object.__setattr__(self, name, value)

    def __delattr__(self, name):
        _lazy_fill(self)
        object.__delattr__(self, name)

    def __reduce_ex__(self, protocol):
        _lazy_fill(self)
        return self.__reduce_ex__(protocol)     # (the object's class is the call class by now)

    raw = None
    for k in cls.__mro__:              # the descriptor that hands out an instance's dict, from whichever base introduced it
        if "__dict__" in vars(k) and hasattr(vars(k)["__dict__"], "__get__"):
            raw = vars(k)["__dict__"].__get__
            break
    if raw is None:
        raise TypeError(f"{cls.__name__} instances have no __dict__")

    def _dict(self):
        _lazy_fill(self)
        return raw(self)
    ns = {"__getattr__": __getattr__, "__setattr__": __setattr__, "__delattr__": __delattr__, "__reduce_ex__": __reduce_ex__,
          "__dict__": property(_dict), "__slots__": (), "_lz_rawdict": staticmethod(raw)}

    def forward(name):
        def get(self):
            _lazy_fill(self)
            return getattr(self, name)
        return get
    # every name the class itself answers (fields with defaults, methods, properties) is shadowed by a data descriptor that fills first;
    # a data descriptor wins over the instance dict, and once filled the object has left this class
    skip = {"__class__", "__dict__", "__weakref__", "__module__", "__doc__", "__slots__", "__new__", "__init__", "__init_subclass__",
            "__subclasshook__", "__getattribute__", "__getattr__", "__setattr__", "__delattr__", "__reduce_ex__", "__reduce__",
            "__sizeof__", "__dir__", "__del__", "__dataclass_fields__", "__dataclass_params__", "__annotations__", "__match_args__",
            "__class_getitem__", "__hash__"}
    for name in dir(cls):
        if name in skip or name in ("qc", "_lz_rawdict"):
            continue
        if name.startswith("__") and name.endswith("__"):
            # special methods are looked up on the type: repr / eq / ordering of a stand-in go through the real object
            if name in ("__repr__", "__str__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__format__", "__getstate__",
                        "__copy__", "__deepcopy__"):
                def special(self, *a, _n=name, **k):
                    _lazy_fill(self)
                    return getattr(self, _n)(*a, **k)
                ns[name] = special
            continue
        ns[name] = property(forward(name), lambda self, v, _n=name: (_lazy_fill(self), object.__setattr__(self, _n, v))[1])
    lz = type("_Lazy" + cls.__name__, (cls,), ns)
    _LAZY_CLASSES[cls] = lz
    return lz


class LazySource:
    """Where the stand-ins of one task's candidate list come from: the record tables - VIEWS of the batch's pinned result block, valid
    until the task's next call on the batch -, the task input and the classes.  At first the candidates of `call_candidates`;
    `set_final` moves it to the records of `finalize_candidates`; `detach` - before the block is handed on: the task's next call, its
    `close()` - turns every stand-in somebody still holds into its call and lets go of the tables (no copy of the records is ever made:
    a consumer that keeps the calls it wants and drops the rest, as `CallTask.execute` does, pays for the calls it keeps)."""

    def __init__(self, res: Result, ti, svcall_cls, bnd_cls, post_cls, batch, keep_all: bool):
        import numpy as np
        self.calls = np.ascontiguousarray(res.calls)
        self.rnames = np.ascontiguousarray(res.rnames, np.uint32)
        self.alt_pool = None
        self.ti, self.cls, self.bnd_cls, self.post_cls, self.batch = ti, svcall_cls, bnd_cls, post_cls, batch
        self.final = False
        self.keep_all = keep_all
        self.stubs = None            # the list call_candidates returned (the stand-ins in record order)
        self.n_filled = 0
        self.n_single = 0            # stand-ins outside the bulk selection that were filled alone
        self.on_detach = None        # called when the tables are let go of (a GPU server's result segment: sniffles_amd.server)

    def make(self, all_qc: bool = False) -> list:
        fast = _load_fast()
        with no_gc():
            self.stubs = fast.make_stubs(lazy_class(self.cls), self, self.calls, 0, len(self.calls), bool(all_qc))
        return list(self.stubs)

    def refresh_qc(self) -> None:
        """`qc` of the records onto the stand-ins (a source that was final from the start: sniffles_amd.server)."""
        if self.stubs is not None:
            _load_fast().stub_refresh_qc(self.stubs, self.calls, 0)

    def set_final(self, res: Result) -> None:
        import numpy as np
        self.calls = np.ascontiguousarray(res.calls)
        self.rnames = np.ascontiguousarray(res.rnames, np.uint32)
        self.alt_pool = np.ascontiguousarray(res.alt_pool, np.uint8)
        self.final = True
        _load_fast().stub_refresh_qc(self.stubs, self.calls, 0)

    def detach(self) -> None:
        """The record tables are about to be handed on: the stand-ins that anything besides this source's own list still refers to
        become calls now; the others are garbage.  Breaks the list <-> source cycle."""
        import numpy as np
        # (No lock around detach / fill: a list of stand-ins belongs to the thread that asked for it.  An RLock taken here - added for two
        # threads touching one list - cost the two-call seam half its speed: 24 x call_candidates / finalize_candidates with two tasks in
        # flight 122-229 ms per genome instead of 66-82 ms, one task at a time 110-120 instead of 99-108 ms, same box, same library, the
        # lock created but not taken: 69-75 ms.  What it waits for was not found; the form before it is restored.)
        if self.stubs is not None and self.calls is not None:
            targets, idx = _load_fast().stub_select(self.stubs, self, 2)
            if targets:
                self.fill_many(targets, np.frombuffer(idx, np.int64))
            del targets
        self.stubs = None
        self.calls = self.rnames = self.alt_pool = None
        cb, self.on_detach = self.on_detach, None
        if cb is not None:
            cb()

    def fill(self, obj) -> None:
        """`obj` (a stand-in of this source) was touched: it and every stand-in the consumer is about to touch become real calls -
        all of them at the candidate stage (`finalize_candidates` then works on objects, as before), the ones with `qc` set (all
        under `no_qc`) after it.  What is left out stays a stand-in and is filled alone if it is ever touched."""
        import numpy as np
        d = _raw_dict(obj)
        me = d.get("_lzi")
        if me is None:                  # (already a call)
            return
        if self.calls is None:
            raise RuntimeError("this call's task was closed (or ran again) while nothing referred to the call")
        if self.stubs is not None and (not self.final or self.keep_all or d.get("qc")):
            targets, idx = _load_fast().stub_select(self.stubs, self, 1 if (self.final and not self.keep_all) else 0)
            self.fill_many(targets, np.frombuffer(idx, np.int64))
        elif self.stubs is not None and self.n_single >= 1:
            # a second call that failed QC is touched: the consumer walks all of them (the SNF writer stores every candidate,
            # parallel.py:278-291) - the rest in one pass, not one by one
            self.keep_all = True
            targets, idx = _load_fast().stub_select(self.stubs, self, 0)
            self.fill_many(targets, np.frombuffer(idx, np.int64))
        else:
            self.n_single += 1
            self.fill_many([obj], np.asarray([me], np.int64))

    def fill_many(self, targets: list, idx) -> None:
        import numpy as np
        fast = _load_fast()
        ti = self.ti
        idx = np.ascontiguousarray(idx, np.int64)
        with no_gc():
            fast.materialize(self.cls, self.bnd_cls, ForwardDifferenceWelford, None if self.final else self.post_cls, self.batch, self.calls,
                             0, len(self.calls), self.rnames, ti.qnames, ti.contig, ti.task_id, ti.contig_names, FILTERS, idx, targets)
            if self.final:
                plain = all(getattr(type(c), "finalize", None) is SVCall.finalize for c in targets)
                fast.apply_final(targets, self.calls, 0, self.alt_pool, ti.ps_names, FILTERS, _QC_SV_EARLY_EXIT, plain, idx)
                if not plain:
                    for c in targets:
                        c.finalize()
        self.n_filled += len(targets)


def lazy_calls_supported(ti, cls=None) -> bool:
    fast = _load_fast()
    if not (fast is not None and hasattr(fast, "make_stubs") and (ti.qnames is None or isinstance(ti.qnames, list))
            and (ti.contig_names is None or isinstance(ti.contig_names, list)) and (ti.ps_names is None or isinstance(ti.ps_names, list))):
        return False
    if cls is not None:                 # a call class whose instances have no `__dict__` (or that cannot be subclassed) is built eagerly
        try:
            lazy_class(cls)
        except TypeError:
            return False
    return True


def is_stand_in(c) -> bool:
    return getattr(type(c), "_lz_rawdict", None) is not None and _raw_dict(c).get("_lz") is not None


def apply_final_py(calls: list, res: Result, ti, lo: int = 0) -> None:
    K, rows = _columns(res, lo, lo + len(calls))
    k_qc, k_filter, k_phs, k_gts, k_vaf, k_al, k_ao = (K[n] for n in ("qc", "filter", "ph_set", "gt_set", "vaf", "alt_len", "alt_off"))
    k_ph = [K[n] for n in ("ph_hp", "ph_ps", "ph_hp_support", "ph_ps_support", "ph_hp_pass", "ph_ps_pass")]
    k_gt = [K[n] for n in ("gt_a", "gt_b", "gt_gq", "gt_dr", "gt_dv", "gt_hp", "gt_ps")]
    pool = res.alt_pool
    for call, r in zip(calls, rows):
        call.qc, call.filter = bool(r[k_qc]), FILTERS[r[k_filter]]
        if call.filter not in _QC_SV_EARLY_EXIT:          # see fill_final
            call.info["COVERAGE_VAR"] = None
        if r[k_phs]:
            hp, ps, hs, pss, hpass, ppass = (r[k] for k in k_ph)
            call.info["PHASE"] = f"{hp},{_ps(ps, ti)},{hs},{pss},{'PASS' if hpass else 'FAIL'},{'PASS' if ppass else 'FAIL'}"
        if r[k_gts]:
            a, b, gq, dr, dv, ghp, gps = (r[k] for k in k_gt)
            call.genotypes[0] = (a, b, gq, dr, dv, (None if ghp < 0 else str(ghp), _ps(gps, ti)))
            call.info["VAF"] = r[k_vaf]
        n = r[k_al]
        if n >= 0:
            o = r[k_ao]
            call.alt = pool[o:o + n].tobytes().decode("latin-1")


def new_call(svcall_cls=SVCall):
    return svcall_cls(contig=None, pos=0, id="", ref="N", alt="", qual=0, filter="PASS", info=dict(), svtype="",
                      svlen=0, end=0, genotypes=dict(), precise=False, support=0, rnames=None, qc=True, nm=-1,
                      postprocess=None)


# ---------------------------------------------------------------------------------------------- multi-sample combine
@dataclass
class SVGroup:
    """Group of per-sample SV calls merged into one multi-sample call (reference `sv.py:226-481`).

    Which candidate joins which group is decided on the GPU (`sniffles_amd.cluster.resolve_block_groups`); this class
    holds the membership, keeps the running means exactly as the reference does, and builds the combined `SVCall`."""
    candidates: list
    pos_mean: float
    len_mean: float
    included_samples: set
    coverages_nonincluded: dict
    bnd_mate_ref_start_mean: float = None
    bnd_mate_contig: str = None

    @classmethod
    def from_candidate(cls, candidate) -> "SVGroup":
        g = cls(candidates=[candidate], pos_mean=float(candidate.pos), len_mean=float(abs(candidate.svlen)),
                included_samples={candidate.sample_internal_id}, coverages_nonincluded=dict())
        if candidate.svtype == "BND":
            g.bnd_mate_contig = candidate.bnd_info.mate_contig
            g.bnd_mate_ref_start_mean = candidate.bnd_info.mate_ref_start
        return g

    def add_candidate(self, candidate) -> None:
        n = len(self.candidates)
        self.pos_mean *= n
        self.len_mean *= n
        self.pos_mean += candidate.pos
        self.len_mean += abs(candidate.svlen)
        bnd = candidate.svtype == "BND"
        if bnd:
            self.bnd_mate_ref_start_mean *= n
            self.bnd_mate_ref_start_mean += candidate.bnd_info.mate_ref_start
        self.candidates.append(candidate)
        n += 1
        self.pos_mean /= n
        self.len_mean /= n
        self.included_samples.add(candidate.sample_internal_id)
        if bnd:
            self.bnd_mate_ref_start_mean /= n

    def call(self, config, task, svcall_cls=None):
        """Combined call of this group or None (reference `SVGroup.call`, sv.py:320-481)."""
        from . import util
        svcall_cls = svcall_cls or SVCall
        first = self.candidates[0]
        n_samples = len(config.snf_input_info)
        single_noqc = config.no_qc and n_samples == 1
        n_pass = sum(c.qc for c in self.candidates)
        n_present = len(self.included_samples)
        confident = (n_pass > 0 and n_pass / float(n_samples) >= config.combine_high_confidence) or \
                    (n_present / float(n_samples) >= config.combine_low_confidence and
                     n_present >= config.combine_low_confidence_abs)
        if not confident and not single_noqc:
            return None
        if not config.combine_output_filtered and not any(c.qc and c.filter == "PASS" for c in self.candidates) \
                and not single_noqc:
            return None
        if getattr(config, "combine_consensus", False):
            raise NotImplementedError("--combine-consensus is broken in the reference (sv.py:382 unpacks 7-tuples into 5)")

        rnames = []
        genotypes = {}
        for c in self.candidates:
            if c.rnames is not None:
                rnames.extend(c.rnames)
            if 0 not in c.genotypes:
                c.genotypes[0] = (".", ".", 0, 0, c.support, (None, None))
            a, b, gq, dr, dv, ps = c.genotypes[0]
            sid = c.sample_internal_id
            if sid in genotypes:  # several calls of one sample in the group: keep the "larger" genotype, chain the ids
                ca, cb, cgq, cdr, cdv, cps, cid = genotypes[sid]
                new_id = cid + "," + config.id_prefix + c.id
                if ca == "." or (a != "." and (a, b) >= (ca, cb)):
                    genotypes[sid] = (a, b, gq, dr, dv, ps, new_id)
                else:
                    genotypes[sid] = (ca, cb, cgq, cdr, cdv, cps, new_id)
            else:
                genotypes[sid] = (a, b, gq, dr, dv, ps, config.id_prefix + c.id)
        for sample in config.snf_input_info:
            sid = sample["internal_id"]
            if sid in genotypes:
                continue
            cov = self.coverages_nonincluded[sid]
            if cov >= config.combine_null_min_coverage:
                genotypes[sid] = (0, 0, 0, cov, 0, (None, None), "NULL")
            else:
                genotypes[sid] = (".", ".", 0, cov, 0, (None, None), "NULL")

        if config.combine_pair_relabel:
            top = (0, 0)
            for a, b, q, *_ in genotypes.values():
                if q > config.combine_pair_relabel_threshold and a != ".":
                    top = max(top, (a, b))
            if top != (0, 0):
                for sid, (a, b, q, dr, dv, ps, nid) in list(genotypes.items()):
                    if q < config.combine_pair_relabel_threshold and a != ".":
                        genotypes[sid] = (top[0], top[1], q, dr, dv, ps, nid)

        med_pos = int(util.median(c.pos for c in self.candidates))
        med_len = int(util.median(c.svlen for c in self.candidates))
        alt = first.alt
        if first.svtype == "INS":
            end = med_pos
            best = abs(len(alt) - med_len)
            for c in self.candidates:
                d = abs(len(c.alt) - med_len)
                if d < best:
                    best, alt = d, c.alt
        else:
            end = med_pos + abs(med_len)
        use_med = getattr(config, "dev_combine_medians", False)

        def avg(attr):
            return util.mean_or_none_round(getattr(c, attr) for c in self.candidates if getattr(c, attr) is not None)

        call = svcall_cls(contig=first.contig, pos=med_pos if use_med else first.pos,
                          id=f"{first.svtype}.{task.sv_id:X}M{task.id:X}", ref="N", alt=alt,
                          qual=util.mean_or_none_round(int(c.qual) for c in self.candidates if c.qual is not None),
                          filter="PASS" if n_samples != 1 else first.filter,
                          info=dict() if n_samples != 1 else first.info, svtype=first.svtype,
                          svlen=med_len if use_med else first.svlen, end=end if use_med else first.end,
                          genotypes=genotypes,
                          precise=sum(int(c.precise) for c in self.candidates) / float(len(self.candidates)) > 0.5,
                          support=round(util.mean(c.support for c in self.candidates)), rnames=rnames, postprocess=None,
                          qc=True, nm=-1)
        call.svlens = None
        call.fwd = sum(c.fwd for c in self.candidates)
        call.rev = sum(c.rev for c in self.candidates)
        for attr in ("coverage_upstream", "coverage_start", "coverage_center", "coverage_end", "coverage_downstream"):
            setattr(call, attr, avg(attr))
        if n_samples != 1:
            call.set_info("STDEV_POS", util.stdev(c.pos for c in self.candidates))
            call.set_info("STDEV_LEN", util.stdev(c.svlen for c in self.candidates))
        if abs(call.svlen) < config.minsvlen_screen:
            return None
        task.sv_id += 1
        return call


def call_groups(svgroups, config, task):
    """Reference `sv.call_groups` (sv.py:642-646)."""
    for g in svgroups:
        c = g.call(config, task)
        if c is not None:
            yield c
