"""Host side of signature extraction: BGZF/BAM container -> inflated alignment records for the GPU.

The reference reads alignments through pysam (`bam.fetch(contig, start, end)`, `leadprov.py:487`); the extraction
kernels (`csrc/snf_extract.hip`) instead take the INFLATED BAM alignment records of one contig as one byte blob plus
record offsets and parse CIGAR / tags / sequence on the GPU.  This module is the container layer in front of that:
BGZF block inflate (zlib), header parse, the record chain (each record starts with its `block_size`), the
overlap filter `fetch` applies, and the interning of read names / contig names to order-preserving ranks
(SoA convention, `sniffles_amd/soa.py`).  Pure I/O and bookkeeping; no signature logic lives here.
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass

import numpy as np


def bgzf_inflate(data: bytes) -> bytes:
    """Concatenated BGZF blocks (gzip members with a BC extra field) -> raw BAM stream."""
    out = []
    p = 0
    n = len(data)
    while p < n:
        if data[p:p + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError(f"not a BGZF block at byte {p}")
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        q = p + 12
        bsize = None
        while q < p + 12 + xlen:
            si1, si2, slen = data[q], data[q + 1], struct.unpack_from("<H", data, q + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", data, q + 4)[0] + 1
            q += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        cdata = data[p + 12 + xlen:p + bsize - 8]
        isize = struct.unpack_from("<I", data, p + bsize - 4)[0]
        raw = zlib.decompress(cdata, -15) if isize else b""
        if len(raw) != isize:
            raise ValueError("BGZF block size mismatch")
        out.append(raw)
        p += bsize
    return b"".join(out)


def bgzf_deflate(raw: bytes, level: int = 1) -> bytes:
    """Raw BAM stream -> BGZF blocks (+ the empty EOF block); the inverse of `bgzf_inflate` (tests, synthetic inputs)."""
    out = []
    for p in list(range(0, len(raw), 0xff00)) + [None]:
        chunk = b"" if p is None else raw[p:p + 0xff00]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = len(cdata) + 25
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + cdata +
                   struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    return b"".join(out)


def bam_stream(ref_names, ref_lens, records, header_text: str = "@HD\tVN:1.6\tSO:coordinate\n") -> bytes:
    """Header + records as an inflated BAM stream (SAM/BAM spec 4.2)."""
    text = header_text.encode("ascii")
    parts = [b"BAM\x01", struct.pack("<i", len(text)), text, struct.pack("<i", len(ref_names))]
    for name, ln in zip(ref_names, ref_lens):
        nm = name.encode("ascii") + b"\0"
        parts += [struct.pack("<i", len(nm)), nm, struct.pack("<i", int(ln))]
    return b"".join(parts) + b"".join(records)


@dataclass
class BamRecords:
    """Inflated alignment records of a BAM stream (all contigs, file order)."""
    ref_names: list          # header reference names, index = BAM refID
    ref_lens: list
    blob: np.ndarray         # uint8, the concatenated records, each starting with its block_size field
    rec_off: np.ndarray      # int64[n+1] byte offset of every record in blob
    ref_id: np.ndarray       # int32[n]
    pos: np.ndarray          # int32[n]

    @property
    def n(self) -> int:
        return int(self.rec_off.shape[0] - 1)

    def qname(self, i: int) -> str:
        o = int(self.rec_off[i])
        l_read_name = int(self.blob[o + 12])
        return bytes(self.blob[o + 36:o + 36 + l_read_name - 1]).decode("ascii")

    def select(self, idx) -> "BamRecords":
        idx = np.asarray(idx, np.int64)
        lens = (self.rec_off[idx + 1] - self.rec_off[idx]).astype(np.int64)
        off = np.zeros(idx.shape[0] + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        # consecutive records are one slice (a contig of a coordinate-sorted BAM is a single run)
        if idx.shape[0]:
            brk = np.nonzero(np.diff(idx) != 1)[0] + 1
            starts = np.r_[0, brk]
            ends = np.r_[brk, idx.shape[0]]
            parts = [self.blob[int(self.rec_off[idx[a]]):int(self.rec_off[idx[b - 1] + 1])] for a, b in zip(starts.tolist(), ends.tolist())]
            blob = np.concatenate(parts) if len(parts) > 1 else parts[0].copy()
        else:
            blob = np.zeros(0, np.uint8)
        return BamRecords(self.ref_names, self.ref_lens, blob, off, self.ref_id[idx].copy(), self.pos[idx].copy())


def parse_bam(raw: bytes) -> BamRecords:
    """Raw (inflated) BAM stream -> header + record table."""
    if raw[:4] != b"BAM\x01":
        raise ValueError("not a BAM stream")
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        names.append(raw[p + 4:p + 4 + l_name - 1].decode("ascii"))
        lens.append(struct.unpack_from("<i", raw, p + 4 + l_name)[0])
        p += 8 + l_name
    start = p
    offs, rid, pos = [0], [], []
    n = len(raw)
    while p < n:
        bs = struct.unpack_from("<i", raw, p)[0]
        if bs < 32 or p + 4 + bs > n:
            raise ValueError(f"truncated BAM record at byte {p}")
        r, ps = struct.unpack_from("<ii", raw, p + 4)
        rid.append(r)
        pos.append(ps)
        p += 4 + bs
        offs.append(p - start)
    blob = np.frombuffer(raw, np.uint8, count=n - start, offset=start).copy()
    return BamRecords(names, lens, blob, np.array(offs, np.int64), np.array(rid, np.int32), np.array(pos, np.int32))


def read_bam(path: str) -> BamRecords:
    with open(path, "rb") as f:
        return parse_bam(bgzf_inflate(f.read()))


def records_from_list(ref_names, ref_lens, records) -> BamRecords:
    """Build the record table from a list of raw record byte strings (each including block_size)."""
    offs = np.zeros(len(records) + 1, np.int64)
    np.cumsum([len(r) for r in records], out=offs[1:])
    blob = np.frombuffer(b"".join(records), np.uint8).copy() if records else np.zeros(0, np.uint8)
    rid = np.array([struct.unpack_from("<i", r, 4)[0] for r in records], np.int32)
    pos = np.array([struct.unpack_from("<i", r, 8)[0] for r in records], np.int32)
    return BamRecords(list(ref_names), list(ref_lens), blob, offs, rid, pos)


def record_flags(recs: BamRecords) -> np.ndarray:
    """FLAG of every record (uint16 at byte 18 of the record, block_size included)."""
    o = recs.rec_off[:-1]
    return recs.blob[o + 18].astype(np.uint16) | (recs.blob[o + 19].astype(np.uint16) << 8)


def contig_records(recs: BamRecords, contig: str) -> BamRecords:
    """The mapped records of one contig in file order (what `bam.fetch(contig)` iterates).  A region fetch
    additionally drops records that do not overlap the region; the kernel applies the reference's own
    `reference_start` window (leadprov.py:500), which is the stricter test, so contig granularity is enough."""
    rid = recs.ref_names.index(contig)
    keep = np.nonzero((recs.ref_id == rid) & ((record_flags(recs) & 0x4) == 0))[0] if recs.n else np.zeros(0, np.int64)
    return recs.select(keep)


def fnv1a64(b: bytes) -> int:
    h = 0xcbf29ce484222325
    for c in b:
        h = ((h ^ c) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def contig_tables(ref_names):
    """Contig names -> (sorted hash table, rank of the hashed name, rank per BAM refID, names in rank order).
    Rank = position in Python str order (SoA convention for mate contigs)."""
    order = sorted(range(len(ref_names)), key=lambda i: ref_names[i])
    rank_of_refid = np.zeros(len(ref_names), np.int32)
    for r, i in enumerate(order):
        rank_of_refid[i] = r
    hashes = np.array([fnv1a64(nm.encode("ascii")) for nm in ref_names], np.uint64)
    if len(set(hashes.tolist())) != len(ref_names):
        raise ValueError("contig name hash collision (or duplicate names) in the BAM header")
    ho = np.argsort(hashes, kind="stable")
    return hashes[ho].copy(), rank_of_refid[ho].copy(), rank_of_refid, [ref_names[i] for i in order]


def qname_ranks(recs: BamRecords):
    """Read names of the records -> (rank per record in Python str order, sorted distinct names).  Names are ASCII
    (SAM spec), so byte order of the NUL-padded names is Python's str order; one `np.unique` over a fixed-width view
    replaces a Python loop over millions of records."""
    n = recs.n
    if n == 0:
        return np.zeros(0, np.uint32), []
    o = recs.rec_off[:-1]
    ln = recs.blob[o + 12].astype(np.int64) - 1                    # l_read_name counts the terminating NUL
    width = int(ln.max()) if n else 0
    if width <= 0:
        return np.zeros(n, np.uint32), [""]
    col = np.arange(width, dtype=np.int64)
    idx = (o + 36)[:, None] + col[None, :]
    mat = np.where(col[None, :] < ln[:, None], recs.blob[np.minimum(idx, recs.blob.shape[0] - 1)], 0).astype(np.uint8)
    keys = np.ascontiguousarray(mat).view(f"S{width}").reshape(n)
    uniq, inv = np.unique(keys, return_inverse=True)
    return inv.astype(np.uint32), [u.decode("ascii") for u in uniq.tolist()]
