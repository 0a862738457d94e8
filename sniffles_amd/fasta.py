"""Reference FASTA access for the two places the path reads reference bases: `LeadProvider._mask_N_coverage`
(`src/sniffles/leadprov.py:420-443`: coverage reads as 0 where the reference base is 'N') and the VCF writer's REF / ALT
resolution (`vcf.py:108-120, 302-342`).  The reference opens `pysam.FastaFile(config.reference)`; pysam is not part of this
package, so this is a plain reader with pysam's `fetch(contig[, start, end]) -> str` contract: an existing `.fai` index is
used (offset / line-bases / line-width arithmetic, one `seek` per fetch), otherwise the file is scanned once to build the same
table in memory.  Plain-text FASTA and gzip (read fully, once) are served; bgzip random access is container I/O this package
leaves to the caller (any object with `fetch` can be handed to `pipeline.call_sample(reference=...)` instead).
Host-side container I/O only: nothing here computes."""
from __future__ import annotations

import gzip
import io
import os


class FastaFile:
    def __init__(self, path: str):
        self.path = path
        self._index = {}          # contig -> (length, offset, line_bases, line_width)
        self._mem = None          # gzip input: the decompressed bytes
        with open(path, "rb") as f:
            gz = f.read(2) == b"\x1f\x8b"
        if gz:
            with gzip.open(path, "rb") as f:
                self._mem = f.read()
            self._scan(io.BytesIO(self._mem))
        else:
            fai = path + ".fai"
            if os.path.exists(fai):
                with open(fai) as f:
                    for line in f:
                        p = line.rstrip("\n").split("\t")
                        if len(p) >= 5:
                            self._index[p[0]] = (int(p[1]), int(p[2]), int(p[3]), int(p[4]))
            else:
                with open(path, "rb") as f:
                    self._scan(f)
        self.references = list(self._index)
        self._handle = None

    def _scan(self, f) -> None:
        name, length, offset, lb, lw, pos = None, 0, 0, 0, 0, 0
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    self._index[name] = (length, offset, lb or 1, lw or 1)
                name = line[1:].split()[0].decode("ascii") if len(line) > 1 and line[1:].split() else ""
                length, lb, lw = 0, 0, 0
                offset = pos + len(line)
            elif name is not None:
                bases = len(line.rstrip(b"\r\n"))
                if lb == 0:
                    lb, lw = bases, len(line)
                length += bases
            pos += len(line)
        if name is not None:
            self._index[name] = (length, offset, lb or 1, lw or 1)

    def get_reference_length(self, contig: str) -> int:
        return self._index[contig][0]

    def fetch(self, contig, start=None, end=None) -> str:
        """pysam semantics: unknown contig -> KeyError; start / end clipped to the contig; start > end -> ValueError."""
        if contig not in self._index:
            raise KeyError(f"sequence '{contig}' not present")
        length, offset, lb, lw = self._index[contig]
        start = 0 if start is None else int(start)
        end = length if end is None else min(int(end), length)
        if start < 0:
            raise ValueError(f"start out of range ({start})")
        if start > end:
            if start >= length:
                return ""
            raise ValueError(f"invalid coordinates: start ({start}) > stop ({end})")
        if start == end:
            return ""
        b0 = offset + (start // lb) * lw + start % lb
        b1 = offset + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        if self._mem is not None:
            raw = self._mem[b0:b1]
        else:
            if self._handle is None:
                self._handle = open(self.path, "rb")
            self._handle.seek(b0)
            raw = self._handle.read(b1 - b0)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("ascii")

    def close(self) -> None:
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
