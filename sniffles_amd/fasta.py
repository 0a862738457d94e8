"""Reference FASTA access for the two places the path reads reference bases: `LeadProvider._mask_N_coverage`
(`src/sniffles/leadprov.py:420-443`: coverage reads as 0 where the reference base is 'N') and the VCF writer's REF / ALT
resolution (`vcf.py:108-120, 302-342`).  The reference opens `pysam.FastaFile(config.reference)`; pysam is not part of this
package, so this is a plain reader with pysam's `fetch(contig[, start, end]) -> str` contract: an existing `.fai` index is
used (offset / line-bases / line-width arithmetic, one `seek` per fetch; for gzip input the offsets index the decompressed text),
otherwise the text is scanned once - one vectorised pass over its newlines - to build the same table in memory.  Plain-text FASTA
and gzip (read fully, once) are served; bgzip random access is container I/O this package
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
        fai = path + ".fai"
        if os.path.exists(fai):       # (offsets of the UNCOMPRESSED text: they index the decompressed buffer of a gzip / bgzip file just as well)
            with open(fai) as f:
                for line in f:
                    p = line.rstrip("\n").split("\t")
                    if len(p) >= 5:
                        self._index[p[0]] = (int(p[1]), int(p[2]), int(p[3]), int(p[4]))
        if gz:
            with gzip.open(path, "rb") as f:
                self._mem = f.read()
            if not self._index:
                self._scan_bytes(self._mem)
        elif not self._index:
            with open(path, "rb") as f:
                self._scan_bytes(f.read())
        self.references = list(self._index)
        self._handle = None

    def _scan_bytes(self, data: bytes) -> None:
        """The `.fai` table of a FASTA text held in memory: one vectorised pass over the newline positions (a human reference has
        ~50 M lines: a Python loop per line took minutes), per record the name, the offset of its first base, the bases and bytes of
        its first sequence line and the number of bases (all sequence lines but the last have the first line's width, as `faidx`
        requires)."""
        import numpy as np
        buf = np.frombuffer(data, np.uint8)
        n = len(buf)
        if n == 0:
            return
        nl = np.flatnonzero(buf == 10)
        starts = np.concatenate(([0], nl + 1))
        starts = starts[starts < n]                               # first byte of every line
        ends = np.concatenate((nl, [n]))[:len(starts)]            # its newline (or the end of the text)
        hdr = np.flatnonzero(buf[starts] == ord(">"))
        for k, h in enumerate(hdr):
            first, last = h + 1, (hdr[k + 1] if k + 1 < len(hdr) else len(starts))      # sequence lines of this record: [first, last)
            head = bytes(data[starts[h] + 1:ends[h]])
            name = head.split()[0].decode("ascii") if head.split() else ""
            if first >= last:
                self._index[name] = (0, int(ends[h]) + 1, 1, 1)
                continue
            ls, le = starts[first:last], ends[first:last]
            cr = (buf[np.maximum(le - 1, ls)] == 13) & (le > ls)  # "\r\n" line ends
            bases = (le - ls) - cr
            lb = int(bases[0]); lw = int((le[0] - ls[0]) + (1 if le[0] < n else 0))
            self._index[name] = (int(bases.sum()), int(ls[0]), lb or 1, lw or 1)

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
