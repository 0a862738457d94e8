"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

pysam is not installed in the build container, so the reference's extraction code (`leadprov.iter_region`,
`read_iterindels`, `read_itersplits`, `Lead.for_bnd`, `sv.classify_splits`) is run against the objects below:
`AlignedSegment` decodes ONE raw BAM alignment record (SAM/BAM spec v1 section 4.2) and exposes exactly the pysam
attributes that code reads, restating pysam 0.22 / htslib semantics:

  reference_start        core.pos
  reference_end          bam_endpos(): pos + sum of M,D,N,=,X lengths (pos+1 when that sum is 0)
  reference_length       reference_end - reference_start
  query_length           core.l_qseq
  query_alignment_start  leading soft clips (hard clips skipped)                       pysam getQueryStart()
  query_alignment_end    l_qseq minus trailing soft clips; with l_qseq == 0: sum of M,I,=,X (+ first leading S)
                                                                                        pysam getQueryEnd()
  query_alignment_length end - start
  query_sequence         4-bit codes -> "=ACMGRSVTWYHKDBN", None when l_qseq == 0
  cigartuples            [(op, len)]          get_tag/has_tag  first tag of that name, integer/float/Z types

**Parity status of this file: pinned only through the reference's own known-answer reads** (`test_bnd_leads.py`:
17 alignments of hg008.bam / hg002.bam whose `Lead.for_bnd` results the reference asserts; they exercise pos,
CIGAR, flags, SA/NM tags, end position and clip lengths).  Everything else follows the published BAM layout.
"""
from __future__ import annotations

import struct

SEQ_CODES = "=ACMGRSVTWYHKDBN"
CONSUMES_REF = (1, 0, 1, 1, 0, 0, 0, 1, 1, 0)   # M I D N S H P = X B
CONSUMES_QRY = (1, 1, 0, 0, 1, 0, 0, 1, 1, 0)


class AlignedSegment:
    def __init__(self, rec: bytes, ref_names):
        (self.block_size, self.reference_id, self.reference_start, l_read_name, self.mapping_quality, _bin,
         n_cigar, self.flag, self.l_qseq, self.next_reference_id, self.next_reference_start,
         self.template_length) = struct.unpack_from("<iiiBBHHHiiii", rec, 0)
        p = 36
        self.query_name = rec[p:p + l_read_name - 1].decode("ascii")
        p += l_read_name
        cig = struct.unpack_from(f"<{n_cigar}I", rec, p)
        self.cigartuples = [(c & 15, c >> 4) for c in cig]
        p += 4 * n_cigar
        self._seq_raw = rec[p:p + (self.l_qseq + 1) // 2]
        p += (self.l_qseq + 1) // 2 + self.l_qseq
        self._tags = self._parse_tags(rec, p, 4 + self.block_size)
        self.reference_name = ref_names[self.reference_id] if self.reference_id >= 0 else None
        self._seq = None

    @staticmethod
    def _parse_tags(rec, p, end):
        tags = []
        fmt = {"A": "c", "c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}
        while p < end:
            tag = rec[p:p + 2].decode("ascii")
            t = chr(rec[p + 2])
            p += 3
            if t in fmt:
                sz = struct.calcsize(fmt[t])
                v = struct.unpack_from("<" + fmt[t], rec, p)[0]
                if t == "A":
                    v = v.decode("ascii")
                p += sz
            elif t in "ZH":
                q = rec.index(b"\0", p)
                v = rec[p:q].decode("ascii")
                p = q + 1
            elif t == "B":
                sub = chr(rec[p])
                cnt = struct.unpack_from("<i", rec, p + 1)[0]
                sz = struct.calcsize(fmt[sub])
                v = list(struct.unpack_from(f"<{cnt}{fmt[sub]}", rec, p + 5))
                p += 5 + cnt * sz
            else:
                raise ValueError(f"unknown aux type {t!r}")
            tags.append((tag, v))
        return tags

    # flags
    @property
    def is_reverse(self): return bool(self.flag & 0x10)
    @property
    def is_secondary(self): return bool(self.flag & 0x100)
    @property
    def is_supplementary(self): return bool(self.flag & 0x800)
    @property
    def is_unmapped(self): return bool(self.flag & 0x4)

    @property
    def reference_end(self):
        if self.is_unmapped or not self.cigartuples:
            return None
        rlen = sum(ln for op, ln in self.cigartuples if CONSUMES_REF[op])
        return self.reference_start + (rlen if rlen else 1)

    @property
    def reference_length(self):
        e = self.reference_end
        return None if e is None else e - self.reference_start

    @property
    def query_length(self): return self.l_qseq

    @property
    def query_alignment_start(self):
        start = 0
        for op, ln in self.cigartuples:
            if op == 5:
                if start != 0 and start != self.l_qseq:
                    raise ValueError("Invalid clipping in CIGAR string")
            elif op == 4:
                start += ln
            else:
                break
        return start

    @property
    def query_alignment_end(self):
        end = self.l_qseq
        if end == 0:
            for op, ln in self.cigartuples:
                if op in (0, 1, 7, 8) or (op == 4 and end == 0):
                    end += ln
        else:
            for op, ln in reversed(self.cigartuples[1:]):
                if op == 5:
                    if end != self.l_qseq:
                        raise ValueError("Invalid clipping in CIGAR string")
                elif op == 4:
                    end -= ln
                else:
                    break
        return end

    @property
    def query_alignment_length(self):
        return self.query_alignment_end - self.query_alignment_start

    @property
    def query_sequence(self):
        if self.l_qseq == 0:
            return None
        if self._seq is None:
            raw = self._seq_raw
            s = []
            for i in range(self.l_qseq):
                b = raw[i >> 1]
                s.append(SEQ_CODES[(b >> 4) if not (i & 1) else (b & 15)])
            self._seq = "".join(s)
        return self._seq

    def has_tag(self, tag): return any(t == tag for t, _ in self._tags)

    def get_tag(self, tag):
        for t, v in self._tags:
            if t == tag:
                return v
        raise KeyError(f"tag '{tag}' not present")


class AlignmentFile:
    """`fetch(contig, start, end)` over an in-memory record table (`sniffles_amd.bam.BamRecords`): the mapped
    records of `contig` overlapping [start, end) in file order, as the BAI/CSI query of htslib returns them."""

    def __init__(self, recs):
        self.recs = recs
        self.references = list(recs.ref_names)

    def get_reference_length(self, contig):
        return self.recs.ref_lens[self.recs.ref_names.index(contig)]

    def segment(self, i):
        o0, o1 = int(self.recs.rec_off[i]), int(self.recs.rec_off[i + 1])
        return AlignedSegment(bytes(self.recs.blob[o0:o1]), self.recs.ref_names)

    def fetch(self, contig=None, start=None, end=None, until_eof=False, region=None):
        rid = self.recs.ref_names.index(contig)
        for i in range(self.recs.n):
            if int(self.recs.ref_id[i]) != rid:
                continue
            seg = self.segment(i)
            if seg.is_unmapped:
                continue
            e = seg.reference_end
            if start is not None and e is not None and e <= start:
                continue
            if end is not None and seg.reference_start >= end:
                continue
            yield seg
