"""The SNF container: reference `src/sniffles/snf.py:29-267`, same class and method names.

An `.snf` file is one JSON header line (`{"config": ..., "index": {contig: {block: [(offset, length), ...]}},
"snf_candidate_count": n}`) followed by the concatenated gzip members of pickled block dicts
`{svtype: [SVCall, ...], "_COVERAGE": {bin_start: depth}}`, one block per `snf_block_size` (100 kb) of a contig.
Files written here are read by the reference and the other way round:

* blocks are pickled under the reference's module path (`sniffles.sv.SVCall`, `SVCallBNDInfo`,
  `ForwardDifferenceWelford`) whichever classes the calls are instances of, and read back into the classes of
  `sniffles_amd.sv`;
* the per-block `_COVERAGE` table (`SNFile.annotate_block_coverages`, snf.py:249-267: the coverage vector averaged per
  `coverage_binsize_combine` bp and rounded) is computed on the GPU from the task's sparse read table
  (`snf_batch_block_coverage`, include/sniffles_amd.h) - the dense vector is never built, and there is no CPU fallback.

Everything else in this module is container I/O (gzip, pickle, JSON, file offsets), like the reference.
"""
from __future__ import annotations

import gzip
import io
import json
import os
import pickle
import sys
import threading
import types
from dataclasses import dataclass
from typing import Optional

from . import sv

REF_MODULE = "sniffles.sv"
_CLASSES = {"SVCall": sv.SVCall, "SVCallBNDInfo": sv.SVCallBNDInfo, "ForwardDifferenceWelford": sv.ForwardDifferenceWelford,
            "SVCallPostprocessingInfo": sv.SVCallPostprocessingInfo}
_LOCK = threading.Lock()


# what a block may name besides the record classes: containers / scalars of the standard library and numpy scalars (the
# reference stores e.g. numpy floats in `nm`).  Anything else in an .snf file is refused instead of imported.
_SAFE_GLOBALS = {
    ("builtins", n) for n in ("set", "frozenset", "dict", "list", "tuple", "int", "float", "complex", "str", "bytes", "bytearray",
                              "bool", "slice", "range", "object")
} | {("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor"), ("copyreg", "__newobj__"),
     ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
     ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")}


class _Unpickler(pickle.Unpickler):
    """Blocks written by the reference name `sniffles.sv.*`; resolve them to this package's record types.  Only those
    classes and a fixed list of harmless globals are accepted: an .snf file cannot make the reader import anything else."""

    def find_class(self, module, name):
        if module in (REF_MODULE, sv.__name__) and name in _CLASSES:
            return _CLASSES[name]
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"SNF block names {module}.{name}: not a Sniffles record class - refusing to load it")


_NAME_OF = {cls: name for name, cls in _CLASSES.items()}
# stand-ins that carry the reference's module path: pickle writes a class as (module, qualname) and checks that the pair
# resolves to the object it was given, so the records are written as instances of these (the record classes themselves are
# never touched) while a module of that name holds them
_PROXIES = {name: type(name, (), {"__module__": REF_MODULE, "__qualname__": name}) for name in _CLASSES}


class _RefPickler(pickle.Pickler):
    def __init__(self, file, classes):
        super().__init__(file)
        self._classes = classes

    def reducer_override(self, obj):
        name = _NAME_OF.get(type(obj))
        if name is None:
            return NotImplemented
        import copyreg
        # (copyreg.__newobj__ insists on the object's own class; _reconstructor is object.__new__(cls) as well)
        return copyreg._reconstructor, (self._classes[name], object, None), obj.__dict__


def _dumps_as_reference(block: dict) -> bytes:
    """pickle.dumps of a block with this package's record classes written under the reference's names.

    When the real `sniffles.sv` is loaded in this process its classes are named directly.  Otherwise the pickler must still
    find, under `sniffles.sv`, the class object it is asked to write: for the duration of the dump a stand-in module of that
    name holds proxy classes (`_PROXIES`).  The record classes of this package are not modified at any time."""
    with _LOCK:
        real = sys.modules.get(REF_MODULE)
        buf = io.BytesIO()
        if real is not None and getattr(real, "__file__", None):
            _RefPickler(buf, {name: getattr(real, name) for name in _CLASSES}).dump(block)
            return buf.getvalue()
        had_pkg = "sniffles" in sys.modules
        try:
            if not had_pkg:
                pkg = types.ModuleType("sniffles")
                pkg.__path__ = []
                sys.modules["sniffles"] = pkg
            mod = types.ModuleType(REF_MODULE)
            for name, proxy in _PROXIES.items():
                setattr(mod, name, proxy)
            sys.modules[REF_MODULE] = mod
            _RefPickler(buf, _PROXIES).dump(block)
            return buf.getvalue()
        finally:
            sys.modules.pop(REF_MODULE, None)
            if not had_pkg:
                sys.modules.pop("sniffles", None)


@dataclass
class SNFPart:
    """What `SNFile.write_results` reads from a task's result (reference `result.py:68-75`, `parallel.py:276-295`)."""
    task_id: int
    contig: str
    snf_filename: str
    snf_index: dict
    snf_total_length: int
    snf_candidate_count: int
    coverage_average_total: float
    has_snf: bool = True


class SNFileBase:
    header_length: int
    _header: Optional[dict]

    def __init__(self, config, handle, filename=None):
        self.config = config
        self.handle = handle
        self.filename = filename
        self.blocks = {}
        self._header = None
        self._index = {}
        self.total_length = 0
        self._results = []

    @classmethod
    def open(cls, filename: str, config=None) -> "SNFileBase":
        if config is None:
            from .config import SnifflesConfig
            config = SnifflesConfig()
        obj = cls(config, open(filename, "rb"), filename)
        obj.read_header()
        return obj

    @property
    def index(self) -> dict:
        return self._index

    @property
    def header(self) -> dict:
        return self._header

    @property
    def population(self):
        return self.header.get("population", None)

    @property
    def reqc(self) -> bool:
        """Was this file written by a version old enough that QC must be redone (snf.py:66-81)?"""
        mode = getattr(self.config, "reqc", "auto")
        if mode == "auto":
            try:
                build, _, _ = self.header["config"]["build"].partition("-")
            except (KeyError, AttributeError, TypeError):
                return True
            return build < "2.5.3"
        return mode

    def is_open(self) -> bool:
        return self.handle is not False

    def _open(self):
        if self.handle is not False:
            self.close()
        self.handle = open(self.filename, "rb")

    def _close_after_use(self):
        if getattr(self.config, "combine_close_handles", False):
            self.close()

    # ---- writing
    def store(self, svcand):
        bs = self.config.snf_block_size
        block_index = int(svcand.pos / bs) * bs
        if block_index not in self.blocks:
            self.blocks[block_index] = {svtype: [] for svtype in sv.TYPES}
            self.blocks[block_index]["_COVERAGE"] = {}
        if not getattr(self.config, "output_rnames", False):
            svcand.rnames = None
        if svcand.svtype in sv.TYPES:
            self.blocks[block_index][svcand.svtype].append(svcand)

    def serialize_block(self, block_id) -> bytes:
        return _dumps_as_reference(self.blocks[block_id])

    def unserialize_block(self, data: bytes):
        return _Unpickler(io.BytesIO(data)).load()

    def write_and_index(self):
        if not self.is_open():
            self._open()
        offset = 0
        for block_id in sorted(self.blocks):
            data = gzip.compress(self.serialize_block(block_id))
            self.handle.write(data)
            self._index[block_id] = (offset, len(data))
            offset += len(data)
            self.total_length += len(data)
        self._close_after_use()

    def add_result(self, result):
        if result.has_snf:
            self._results.append(result)

    def _calculate_contig_coverages(self, contigs) -> dict:
        per = {c: [] for c in contigs}
        for r in self._results:
            per[r.contig].append(r.coverage_average_total)
        return {c: (sum(v) / len(v) if len(v) > 0 else 0) for c, v in per.items()}

    def _create_header(self, config, main_index: dict, snf_candidate_count: int) -> dict:
        return {"config": config.__dict__, "index": main_index, "snf_candidate_count": snf_candidate_count}

    def write_results(self, config, contigs) -> int:
        """Concatenate the per-task part files behind one header; returns the candidate count (snf.py:186-223)."""
        main_index = {}
        offset = 0
        snf_candidate_count = sum(r.snf_candidate_count for r in self._results)
        parts_sorted = sorted(self._results, key=lambda r: r.task_id)
        for part in parts_sorted:
            idx = main_index.setdefault(part.contig, {})
            for block, (start, length) in part.snf_index.items():
                idx.setdefault(block, []).append((start + offset, length))
            offset += part.snf_total_length
        config.contig_coverages = self._calculate_contig_coverages(contigs)
        header = self._create_header(config, main_index, snf_candidate_count)
        self.handle.write((json.dumps(header, default=lambda obj: "<Unstored_Object>") + "\n").encode())
        for part in parts_sorted:
            with open(part.snf_filename, "rb") as h:
                self.handle.write(h.read())
            os.remove(part.snf_filename)
        return snf_candidate_count

    # ---- reading
    def read_header(self):
        if not self.is_open():
            self._open()
        header_text = self.handle.readline()
        self.header_length = len(header_text)
        try:
            self._header = json.loads(header_text.strip())
        except Exception as e:
            raise ValueError(f"'{self.filename}' is not a valid .snf file (header: {e})") from e
        self._index = self._header["index"]
        self._close_after_use()

    def read_blocks(self, contig, block_index):
        if not self.is_open():
            self._open()
        block_index = str(block_index)
        if contig not in self.index or block_index not in self.index[contig]:
            self._close_after_use()
            return None
        blocks = []
        try:
            for start, length in self.index[contig][block_index]:
                self.handle.seek(self.header_length + start)
                blocks.append(self.unserialize_block(gzip.decompress(self.handle.read(length))))
        finally:
            self._close_after_use()
        return blocks

    def get_index(self):
        return self.index

    def get_total_length(self):
        return self.total_length

    def close(self) -> None:
        if self.handle is not False:
            self.handle.close()
            self.handle = False

    def get_all_blocks(self, contig: str) -> dict:
        blocks = {}
        if contig in self.index:
            for block_start in self.index[contig].keys():
                blocks[block_start] = self.read_blocks(contig, block_start)[0]
        return blocks

    def get_full_coverage(self, contig: str) -> dict:
        coverage = {}
        for b in self.get_all_blocks(contig).values():
            coverage.update(b["_COVERAGE"])
        return coverage


class SNFile(SNFileBase):
    def annotate_block_coverages(self, lead_provider):
        """Downsampled coverage of every stored block (snf.py:249-267).  `lead_provider` is the task's
        `sniffles_amd.leadprov.LeadProvider` after `Task.call_candidates` (or the `Task` itself): its read table is in
        HBM with the ends sorted, and one launch per task evaluates all bins of all blocks."""
        batch = getattr(lead_provider, "device_batch", None)
        if batch is None:
            batch = getattr(lead_provider, "_batch", None)
        if batch is None:
            raise RuntimeError("annotate_block_coverages needs the task's device batch: call Task.call_candidates first "
                               "(the coverage lives on the GPU; there is no CPU fallback)")
        binsize = self.config.coverage_binsize_combine
        block_size = self.config.snf_block_size
        per_block = block_size // binsize
        want = []   # (block_offset, bin position key, index into the downsampled vector)
        for block_offset in self.blocks.keys():
            block_index = block_offset // block_size
            for i in range(per_block):
                want.append((block_offset, block_offset + i * binsize, block_index * per_block + i))
        if not want:
            return
        lo = min(w[2] for w in want)
        hi = max(w[2] for w in want) + 1
        depth = batch.block_coverage(0, binsize, lo, hi - lo)
        for block_offset, key, j in want:
            d = int(depth[j - lo])
            if d >= 0:      # beyond the padded vector: the reference's IndexError, the bin stays unset
                self.blocks[block_offset]["_COVERAGE"][key] = d
