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


# ---------------------------------------------------------------------------------------------- container layout
# A file is  <header line> <member> <member> ...  where a member is one gzip stream of one pickled block.  Offsets in the
# header's index are relative to the first byte after the header line.  A part file (one per task, written by a worker) is
# the members alone; `write_results` strings the parts together in task order and shifts their offsets.

def _block_start(pos: int, block_size: int) -> int:
    return int(pos / block_size) * block_size


def _empty_block() -> dict:
    block = {svtype: [] for svtype in sv.TYPES}
    block["_COVERAGE"] = {}
    return block


def _header_line(config, index: dict, candidate_count: int) -> bytes:
    doc = {"config": config.__dict__, "index": index, "snf_candidate_count": candidate_count}
    return (json.dumps(doc, default=lambda obj: "<Unstored_Object>") + "\n").encode()


def _shifted_index(parts) -> dict:
    """{contig: {block: [(offset, length), ...]}} of the parts laid end to end in the given order."""
    merged, base = {}, 0
    for part in parts:
        per_contig = merged.setdefault(part.contig, {})
        for block, (start, length) in part.snf_index.items():
            per_contig.setdefault(block, []).append((start + base, length))
        base += part.snf_total_length
    return merged


class _Handle:
    """The file handle of a container with the reference's conventions: `False` means closed, a reader re-opens by name on
    demand, and with `--combine-close-handles` every operation closes the file behind itself (hundreds of samples)."""

    def __init__(self, handle, filename, close_after_use: bool):
        self.handle, self.filename, self.close_after_use = handle, filename, close_after_use

    @property
    def is_open(self) -> bool:
        return self.handle is not False

    def need(self):
        if not self.is_open:
            self.handle = open(self.filename, "rb")
        return self.handle

    def close(self) -> None:
        if self.is_open:
            self.handle.close()
            self.handle = False

    def done(self) -> None:
        if self.close_after_use:
            self.close()


class SNFileBase:
    """Reference interface (`snf.py:29-241`: same method names and on-disk format) over the pieces above."""

    def __init__(self, config, handle, filename=None):
        self.config = config
        self.filename = filename
        self._io = _Handle(handle, filename, bool(getattr(config, "combine_close_handles", False)))
        self.blocks = {}            # block start -> {svtype: [SVCall], "_COVERAGE": {bin: depth}}   (writing)
        self._header = None         # parsed header line                                               (reading)
        self.header_length = 0
        self._index = {}            # writing: block -> (offset, length) of this part; reading: the header's index
        self.total_length = 0
        self._results = []          # SNFPart of every task that produced a part

    @classmethod
    def open(cls, filename: str, config=None) -> "SNFileBase":
        if config is None:
            from .config import SnifflesConfig
            config = SnifflesConfig()
        obj = cls(config, open(filename, "rb"), filename)
        obj.read_header()
        return obj

    # the reference exposes the raw handle (`False` once closed)
    @property
    def handle(self):
        return self._io.handle

    @handle.setter
    def handle(self, value):
        self._io.handle = value

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
        if mode != "auto":
            return mode
        try:
            build = self.header["config"]["build"].partition("-")[0]
        except (KeyError, AttributeError, TypeError):
            return True
        return build < "2.5.3"

    def is_open(self) -> bool:
        return self._io.is_open

    def close(self) -> None:
        self._io.close()

    def get_index(self):
        return self.index

    def get_total_length(self):
        return self.total_length

    # ---- writing a part
    def store(self, svcand):
        start = _block_start(svcand.pos, self.config.snf_block_size)
        block = self.blocks.get(start)
        if block is None:
            block = self.blocks[start] = _empty_block()
        if not getattr(self.config, "output_rnames", False):
            svcand.rnames = None
        if svcand.svtype in sv.TYPES:
            block[svcand.svtype].append(svcand)

    def serialize_block(self, block_id) -> bytes:
        return _dumps_as_reference(self.blocks[block_id])

    def unserialize_block(self, data: bytes):
        return _Unpickler(io.BytesIO(data)).load()

    def write_and_index(self):
        out = self._io.need()
        for block_id in sorted(self.blocks):
            member = gzip.compress(self.serialize_block(block_id))
            out.write(member)
            self._index[block_id] = (self.total_length, len(member))
            self.total_length += len(member)
        self._io.done()

    # ---- writing the final file from the parts
    def add_result(self, result):
        if result.has_snf:
            self._results.append(result)

    def _calculate_contig_coverages(self, contigs) -> dict:
        means = {}
        for c in contigs:
            values = [r.coverage_average_total for r in self._results if r.contig == c]
            means[c] = sum(values) / len(values) if values else 0
        return means

    def write_results(self, config, contigs) -> int:
        """Header, then the part files in task order (each is removed); returns the candidate count (snf.py:186-223)."""
        parts = sorted(self._results, key=lambda r: r.task_id)
        count = sum(r.snf_candidate_count for r in parts)
        config.contig_coverages = self._calculate_contig_coverages(contigs)
        self.handle.write(_header_line(config, _shifted_index(parts), count))
        for part in parts:
            with open(part.snf_filename, "rb") as h:
                self.handle.write(h.read())
            os.remove(part.snf_filename)
        return count

    # ---- reading
    def read_header(self):
        line = self._io.need().readline()
        self.header_length = len(line)
        try:
            self._header = json.loads(line.strip())
        except Exception as e:
            raise ValueError(f"'{self.filename}' is not a valid .snf file (header: {e})") from e
        self._index = self._header["index"]
        self._io.done()

    def read_blocks(self, contig, block_index):
        """The blocks stored for (contig, block start) - one per task that wrote into it - or None."""
        members = self.index.get(contig, {}).get(str(block_index))
        if members is None:
            self._io.need()            # (the reference opens the file before it looks)
            self._io.done()
            return None
        f = self._io.need()
        try:
            blocks = []
            for start, length in members:
                f.seek(self.header_length + start)
                blocks.append(self.unserialize_block(gzip.decompress(f.read(length))))
            return blocks
        finally:
            self._io.done()

    def block_starts(self, contig) -> list:
        """Block starts stored for `contig` (what `candstore.ContigColumns` walks once to keep the reader's candidates as columns)."""
        return [int(k) for k in self.index.get(contig, {})]

    def get_all_blocks(self, contig: str) -> dict:
        return {start: self.read_blocks(contig, start)[0] for start in self.index.get(contig, {})}

    def get_full_coverage(self, contig: str) -> dict:
        coverage = {}
        for block in self.get_all_blocks(contig).values():
            coverage.update(block["_COVERAGE"])
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
