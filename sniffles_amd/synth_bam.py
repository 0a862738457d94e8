"""Seeded synthetic BAM alignment records for the extraction path (tests + tools/bench_extract.py).

No BAM of realistic size exists in this repository or on the GPU box, so the records are synthesised: raw
little-endian alignment records exactly as they appear in an inflated BAM stream (SAM/BAM spec 4.2), with the
features the reference's extraction reads (`leadprov.py:474-655`, `sv.py:649-782`): CIGARs with all nine
operations, leading / trailing soft and hard clips (short, "single-break" sized and long-INS sized), large and
small indels, flags (reverse, secondary, supplementary, duplicate), MAPQ, NM / HP / PS in several integer
encodings, SA strings with 1..n entries on the same and on other contigs / strands, and unrelated tags of every
aux type (A c C s S i I f Z H B) interleaved so that the tag walk is exercised.
"""
from __future__ import annotations

import struct

import numpy as np

OPS = "MIDNSHP=X"
M, I, D, N, S, H, P, EQ, X = range(9)


def _int_tag(tag: str, v: int, rng) -> bytes:
    """Smallest-fitting integer encodings at random, like htslib writers differ."""
    cands = []
    if -128 <= v <= 127: cands.append(("c", "b"))
    if 0 <= v <= 255: cands.append(("C", "B"))
    if -32768 <= v <= 32767: cands.append(("s", "h"))
    if 0 <= v <= 65535: cands.append(("S", "H"))
    if -2 ** 31 <= v < 2 ** 31: cands.append(("i", "i"))
    if 0 <= v < 2 ** 32: cands.append(("I", "I"))
    t, f = cands[int(rng.integers(len(cands)))]
    return tag.encode() + t.encode() + struct.pack("<" + f, v)


def _junk_tag(rng) -> bytes:
    k = int(rng.integers(8))
    name = bytes([int(rng.integers(97, 123)), int(rng.integers(48, 58))])   # lowercase+digit: never NM/HP/PS/SA
    if k == 0: return name + b"A" + bytes([int(rng.integers(65, 91))])
    if k == 1: return name + b"f" + struct.pack("<f", float(rng.random()))
    if k == 2: return name + b"Z" + bytes(rng.integers(33, 127, int(rng.integers(0, 200))).astype(np.uint8)) + b"\0"
    if k == 3: return name + b"H" + b"1AE301" + b"\0"
    if k == 4:
        sub, f = [("c", "b"), ("C", "B"), ("s", "h"), ("S", "H"), ("i", "i"), ("I", "I"), ("f", "f")][int(rng.integers(7))]
        cnt = int(rng.integers(0, 70))
        vals = [1.5] * cnt if sub == "f" else [int(x) for x in rng.integers(0, 100, cnt)]
        return name + b"B" + sub.encode() + struct.pack("<i", cnt) + struct.pack(f"<{cnt}{f}", *vals)
    return _int_tag(name.decode(), int(rng.integers(-1000, 70000)), rng)


def cigar_string(ops) -> str:
    return "".join(f"{ln}{OPS[op]}" for op, ln in ops)


def make_record(ref_id, pos, mapq, flag, qname, ops, seq_codes, tags: bytes) -> bytes:
    l_seq = int(len(seq_codes))
    name = qname.encode("ascii") + b"\0"
    cig = struct.pack(f"<{len(ops)}I", *[(ln << 4) | op for op, ln in ops])
    sc = np.asarray(seq_codes, np.uint8)
    if l_seq & 1:
        sc = np.concatenate([sc, np.zeros(1, np.uint8)])
    packed = ((sc[0::2] << 4) | sc[1::2]).astype(np.uint8).tobytes()
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name), mapq, 4680, len(ops), flag, l_seq, -1, -1, 0)
    body += name + cig + packed + b"\xff" * l_seq + tags
    return struct.pack("<i", len(body)) + body


def _body_ops(rng, ref_span, style):
    """Aligned part of a CIGAR covering about ref_span reference bases."""
    ops = []
    left = ref_span
    big_p = {"fuzz": 0.08, "ont": 0.0006}[style]
    while left > 0:
        run = int(min(left, max(1, rng.exponential({"fuzz": 60, "ont": 14}[style]))))
        ops.append(([M, EQ, X][int(rng.integers(3))] if style == "fuzz" and rng.random() < 0.3 else M, run))
        left -= run
        if left <= 0:
            break
        r = rng.random()
        if r < big_p:                        # SV-sized event
            ln = int(rng.choice([44, 45, 46, 60, 150, 320, 1200, 6000])) + int(rng.integers(0, 3))
            ops.append((I if rng.random() < 0.5 else D, ln))
            if ops[-1][0] == D: left -= ln
        elif r < big_p + 0.01 and style == "fuzz":
            ops.append((N, int(rng.integers(1, 500)))); left -= ops[-1][1]
        elif r < big_p + 0.015 and style == "fuzz":
            ops.append((P, int(rng.integers(1, 5))))
        elif r < big_p + 0.02 and style == "fuzz":
            ops.append((I if rng.random() < 0.5 else D, int(rng.integers(8, 14))))   # around the NM "large" limit 10
            if ops[-1][0] == D: left -= ops[-1][1]
        else:
            ln = int(rng.integers(1, 4))
            ops.append((I if rng.random() < 0.5 else D, ln))
            if ops[-1][0] == D: left -= ln
    if ops[-1][0] not in (M, EQ, X):
        ops.append((M, int(rng.integers(1, 30))))
    return ops


def _clip(rng, style, hard):
    r = rng.random()
    if r < 0.35: return []
    op = H if hard else S
    if r < 0.6: return [(op, int(rng.integers(1, 44)))]
    if r < 0.8: return [(op, int(rng.integers(44, 200)))]
    if r < 0.9: return [(op, int(rng.integers(1240, 1260)))]
    return [(op, int(rng.integers(1260, 9000)))]


def _sa_entry(rng, ref_names, contig, pos, ref_end, q0, q1, qlen, rev, far=False):
    """One SA element placed so that classify_splits sees INS / DEL / DUP / INV / other-contig geometry."""
    kind = int(rng.integers(7))
    name = contig if kind < 5 and not far else ref_names[int(rng.integers(len(ref_names)))]
    strand_rev = rev if kind < 4 else (not rev if kind == 4 else bool(rng.integers(2)))
    span = int(rng.integers(30, 3000))
    readspan = max(1, span + int(rng.integers(-20, 21)))
    before = bool(rng.integers(2))
    gap_q = int(rng.choice([0, 10, 44, 45, 50, 300, 3000]))
    gap_r = int(rng.choice([-500, -50, 0, 10, 44, 45, 50, 300, 5000]))
    if before:
        lead = max(0, q0 - gap_q - readspan)
        spos = max(0, pos - gap_r - span)
    else:
        lead = q1 + gap_q
        spos = max(0, ref_end + gap_r)
    tail = max(0, qlen - lead - readspan)
    parts = []
    c0, c1 = (tail, lead) if strand_rev else (lead, tail)
    if c0: parts.append(f"{c0}{'SH'[int(rng.integers(2))]}")
    if rng.random() < 0.3 and span > 10:
        a = span // 2
        parts += [f"{a}M", f"{int(rng.integers(1, 60))}{'ID'[int(rng.integers(2))]}", f"{span - a}M"]
    else:
        parts.append(f"{span}M")
    if c1: parts.append(f"{c1}{'SH'[int(rng.integers(2))]}")
    cigar = "".join(parts)
    if rng.random() < 0.01:
        cigar = cigar.replace("M", "P", 1)            # malformed for CIGAR_analyze: exception path
    return f"{name},{spos + 1},{'-' if strand_rev else '+'},{cigar},{int(rng.choice([0, 19, 20, 60, 60, 60, 60]))},{int(rng.integers(0, 500))}"


def gen_records(seed: int, n_reads: int, ref_names=("chrA", "chr10", "chr2", "chrB_alt"),
                ref_lens=(400000, 300000, 300000, 100000), contig_index=0, style="fuzz",
                read_len_mean=3000, phased=0.5, sa_frac=0.25, with_tags=True):
    """Records of one contig (+ a few on others), position-sorted like a coordinate-sorted BAM.
    Returns (ref_names, ref_lens, [record bytes])."""
    rng = np.random.default_rng(seed)
    ref_names, ref_lens = list(ref_names), list(ref_lens)
    clen = ref_lens[contig_index]
    contig = ref_names[contig_index]
    recs = []
    for r in range(n_reads):
        ref_span = int(max(50, min(clen // 2, rng.exponential(read_len_mean))))
        pos = int(rng.integers(0, clen - ref_span))
        rev = bool(rng.integers(2))
        supp = rng.random() < 0.15
        flag = (0x10 if rev else 0) | (0x800 if supp else 0)
        if rng.random() < 0.04: flag |= 0x100
        if rng.random() < 0.04: flag |= 0x400
        if rng.random() < 0.02 and style == "fuzz": flag |= 0x200
        mapq = int(rng.choice([0, 5, 19, 20, 21, 60, 60, 60, 60]))
        ops = _clip(rng, style, supp and rng.random() < 0.8) + _body_ops(rng, ref_span, style) + \
            _clip(rng, style, supp and rng.random() < 0.8)
        if style == "fuzz" and rng.random() < 0.03:
            ops = [(H, 5)] + [o for o in ops if o[0] != H] + [(H, 7)]     # H outside S
            ops = [ops[0], (S, 60)] + ops[1:-1] + [(S, 50), ops[-1]]
        qlen = sum(ln for op, ln in ops if op in (M, I, S, EQ, X))
        ref_end = pos + sum(ln for op, ln in ops if op in (M, D, N, EQ, X))
        q0 = sum(ln for op, ln in ops[:2] if op == S)
        q1 = qlen - sum(ln for op, ln in ops[-2:] if op == S)
        seq = rng.integers(0, 4, qlen)
        seq = np.array([1, 2, 4, 8], np.uint8)[seq]
        if qlen and rng.random() < 0.2:
            seq[rng.integers(0, qlen, max(1, qlen // 200))] = 15
        qname = f"r{seed}_{r // 2 if supp else r}_{int(rng.integers(1000))}" if style == "fuzz" else f"read{seed}_{r}"
        tags = []
        if with_tags:
            if rng.random() < 0.9: tags.append(_int_tag("NM", int(rng.integers(0, 4000)), rng))
            if rng.random() < phased:
                tags.append(_int_tag("HP", int(rng.integers(1, 3)), rng))
                if rng.random() < 0.9: tags.append(_int_tag("PS", int(rng.choice([5, 17, 100, 99, 1000001, 23])), rng))
            if rng.random() < sa_frac:
                n_sa = int(rng.choice([1, 1, 1, 1, 2, 2, 2, 3, 3, 5, 9]))
                q_rev0, q_rev1 = (qlen - q1, qlen - q0) if rev else (q0, q1)
                sa = ";".join(_sa_entry(rng, ref_names, contig, pos, ref_end, q_rev0, q_rev1, qlen, rev,
                                        far=rng.random() < 0.2) for _ in range(n_sa)) + ";"
                tags.append(b"SAZ" + sa.encode() + b"\0")
            for _ in range(int(rng.integers(0, 5))):
                tags.insert(int(rng.integers(len(tags) + 1)), _junk_tag(rng))
        recs.append((pos, make_record(contig_index, pos, mapq, flag, qname, ops, seq, b"".join(tags))))
    recs.sort(key=lambda t: t[0])
    return ref_names, ref_lens, [b for _, b in recs]


def gen_sample(seed: int, ref_names=("chr20", "chrM_short", "chr21"), ref_lens=(1_400_000, 16_000, 1_100_000), cov=20.0,
               read_len_mean=12000, site_spacing=18000, err=0.03, split_spacing=0, tr_frac=0.0, site_seed=None):
    """A coherent synthetic sample: SV sites shared by the reads that cross them, so that clusters and calls form.

    Per contig: INS / DEL sites every ~`site_spacing` bp (lengths 50 ... 3000, 40 % homozygous, the rest on one
    haplotype), reads with uniform starts and exponential lengths on haplotype 1 or 2 (80 % carry HP / PS tags),
    the site's allele in the CIGAR of every read that spans it on a carrying haplotype (length jitter 1.5 %, position
    jitter 2 bp, inserted bases = the site's allele with `err` substitutions), 1-2 bp alignment noise in between,
    occasional sub-threshold (45-49 bp) noise events, NM tags, MAPQ 60.  Coordinate-sorted like a BAM.
    `split_spacing` > 0 adds a split-alignment event every ~that many bp, cycling through large deletion, tandem
    duplication, inversion breakpoint and translocation to the next long contig: every carrying read that crosses the
    breakpoint becomes a primary record (soft-clipped, SA tag) plus a supplementary record (0x800, hard- or soft-clipped,
    SA tag back to the primary) - the geometry `classify_splits` / `Lead.for_bnd` turn into DEL / DUP / INV / BND leads.
    `tr_frac` > 0 makes that fraction of the sites tandem-repeat-like (breakpoint scatter of 40 bp instead of 2) and
    returns the annotation as a fourth value: {contig: [(start, end), ...]} padded by 500 bp like `util.load_tandem_repeats`.
    `site_seed`: draw the SV sites from their own generator, so that several samples (different `seed`) share the sites
    of one population; a site is then present in a sample with probability 0.7.
    Returns (ref_names, ref_lens, [record bytes])."""
    rng = np.random.default_rng(seed)
    read_rng = rng
    if site_seed is not None:
        rng = np.random.default_rng(site_seed)       # sites of the population; the reads keep this sample's generator
    code = np.array([1, 2, 4, 8], np.uint8)
    recs = []
    tandem = {}
    for cid, (name, clen) in enumerate(zip(ref_names, ref_lens)):
        sites = []
        p = int(rng.integers(4000, site_spacing))
        while p < clen - 6000:
            ln = int(rng.choice([50 + int(rng.exponential(150)), int(rng.normal(320, 15)), int(rng.normal(2800, 60))], p=[0.7, 0.2, 0.1]))
            ln = max(50, ln)
            ins = bool(rng.integers(2))
            sites.append(dict(pos=p, ln=ln, ins=ins, hap=0 if rng.random() < 0.4 else int(rng.integers(1, 3)),
                              allele=rng.integers(0, 4, ln if ins else 0), tr=bool(tr_frac) and rng.random() < tr_frac))
            p += ln + int(rng.integers(site_spacing // 2, site_spacing * 3 // 2))
        tandem[name] = [(max(0, x["pos"] - 150 - 500), x["pos"] + 150 + 500) for x in sites if x["tr"]]
        site_rng = rng
        if site_seed is not None:
            sites = [x for x in sites if read_rng.random() < 0.7]
            rng = read_rng
        splits = []
        if split_spacing:
            long_ids = [i for i, n in enumerate(ref_lens) if n >= 1_000_000 and i != cid]
            bp, kind = int(split_spacing // 2), 0
            while bp < clen - 30000:
                ev = dict(bp=bp, kind=("DEL", "DUP", "INV", "BND")[kind % 4], ln=int(rng.integers(3000, 15000)),
                          hap=0 if rng.random() < 0.5 else int(rng.integers(1, 3)))
                if ev["kind"] == "BND":
                    if long_ids:
                        ev["cid2"] = long_ids[int(rng.integers(len(long_ids)))]
                        ev["pos2"] = int(rng.integers(20000, ref_lens[ev["cid2"]] - 40000))
                    else:
                        ev["kind"] = "DEL"
                splits.append(ev)
                bp += int(rng.integers(split_spacing * 3 // 4, split_spacing * 5 // 4)); kind += 1
        n_reads = int(cov * clen / read_len_mean)
        for r in range(n_reads):
            span = int(max(1500, min(clen // 3, rng.exponential(read_len_mean))))
            pos = int(rng.integers(0, max(1, clen - span)))
            hap = int(rng.integers(1, 3))
            rev = bool(rng.integers(2))
            ev = next((e for e in splits if pos + 600 <= e["bp"] <= pos + span - 600 and e["hap"] in (0, hap)), None)
            if ev is not None:
                # two-part split read: query [0, a) on the left of the breakpoint, [a, span) at the event's other side
                a, b = ev["bp"] - pos, span - (ev["bp"] - pos)
                jit = int(rng.integers(-2, 3))
                a, b = a + jit, b - jit
                k = ev["kind"]
                cid2 = ev.get("cid2", cid)
                if k == "DEL": start2, rev2 = ev["bp"] + jit + ev["ln"], rev
                elif k == "DUP": start2, rev2 = max(0, ev["bp"] + jit - ev["ln"]), rev
                elif k == "INV": start2, rev2 = ev["bp"] + ev["ln"] - b, not rev
                else: start2, rev2 = ev["pos2"] + jit, rev
                start2 = int(min(max(0, start2), ref_lens[cid2] - b - 1))
                hard = rng.random() < 0.5
                ops1 = [(M, a), (S, b)]
                ops2 = ([(M, b), (H if hard else S, a)] if rev2 != rev else [(H if hard else S, a), (M, b)])
                sa_ops2 = [(S if op == H else op, ln) for op, ln in ops2]
                nm1, nm2 = int(rng.integers(0, a // 50 + 1)), int(rng.integers(0, b // 50 + 1))
                qname = f"s{seed}_{name}_{r}"
                seq_full = code[rng.integers(0, 4, span)]
                phase = [_int_tag("HP", hap, rng), _int_tag("PS", 1000 * (cid + 1) + 7, rng)] if rng.random() < 0.8 else []
                sa1 = f"{ref_names[cid2]},{start2 + 1},{'-' if rev2 else '+'},{cigar_string(sa_ops2)},60,{nm2};"
                sa2 = f"{name},{pos + 1},{'-' if rev else '+'},{cigar_string(ops1)},60,{nm1};"
                recs.append((cid, pos, make_record(cid, pos, 60, 0x10 if rev else 0, qname, ops1, seq_full,
                                                   b"".join([_int_tag("NM", nm1, rng)] + phase + [b"SAZ" + sa1.encode() + b"\0"]))))
                seq2 = seq_full if not hard else (seq_full[:b] if rev2 != rev else seq_full[a:])
                recs.append((cid2, start2, make_record(cid2, start2, 60, 0x800 | (0x10 if rev2 else 0), qname, ops2, seq2,
                                                       b"".join([_int_tag("NM", nm2, rng)] + phase + [b"SAZ" + sa2.encode() + b"\0"]))))
                continue
            ops, seq_parts, nm = [], [], 0
            cur, end = pos, min(clen, pos + span)

            def aligned(upto):
                nonlocal cur, nm
                while cur < upto:
                    run = int(min(upto - cur, max(1, rng.exponential(250))))
                    ops.append((M, run)); seq_parts.append(rng.integers(0, 4, run)); cur += run
                    if cur < upto and rng.random() < 0.7:
                        k = int(rng.integers(1, 3))
                        if rng.random() < 0.5:
                            ops.append((I, k)); seq_parts.append(rng.integers(0, 4, k))
                        elif cur + k < upto:
                            ops.append((D, k)); cur += k
                        nm += k
            for s in sites:
                if s["pos"] <= pos + 300 or s["pos"] + (0 if s["ins"] else s["ln"]) >= end - 300:
                    continue
                if s["hap"] not in (0, hap):
                    continue
                at = s["pos"] + int(round(rng.normal(0, 40 if s["tr"] else 2)))
                if at <= cur + 10:
                    continue
                aligned(at)
                ln = max(45, int(round(s["ln"] * (1 + rng.normal(0, 0.015)))))
                if s["ins"]:
                    a = np.resize(s["allele"], ln).copy()
                    flip = rng.random(ln) < err
                    a[flip] = rng.integers(0, 4, int(flip.sum()))
                    ops.append((I, ln)); seq_parts.append(a)
                else:
                    ops.append((D, ln)); cur += ln
                nm += ln
                ops.append((M, 40)); seq_parts.append(rng.integers(0, 4, 40)); cur += 40
            if rng.random() < 0.05 and end - cur > 2000:        # a sub-threshold noise event
                aligned(cur + int(rng.integers(500, 1500)))
                k = int(rng.integers(45, 50))
                ops.append((D, k)); cur += k; nm += k
                ops.append((M, 30)); seq_parts.append(rng.integers(0, 4, 30)); cur += 30
            aligned(max(end, cur + 50))
            merged = []
            for op, ln in ops:                                    # adjacent M runs are one operation
                if merged and merged[-1][0] == op == M:
                    merged[-1] = (M, merged[-1][1] + ln)
                else:
                    merged.append((op, ln))
            seq = code[np.concatenate(seq_parts)]
            tags = [_int_tag("NM", nm + int(rng.integers(0, span // 60)), rng)]
            if rng.random() < 0.8:
                tags += [_int_tag("HP", hap, rng), _int_tag("PS", 1000 * (cid + 1) + 7, rng)]
            recs.append((cid, pos, make_record(cid, pos, 60, 0x10 if rev else 0, f"s{seed}_{name}_{r}", merged, seq, b"".join(tags))))
        rng = site_rng
    recs.sort(key=lambda t: (t[0], t[1]))
    if tr_frac:
        return list(ref_names), list(ref_lens), [b for _, _, b in recs], tandem
    return list(ref_names), list(ref_lens), [b for _, _, b in recs]
