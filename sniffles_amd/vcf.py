"""VCF output: the writer side of the reference's `VCF` class (`src/sniffles/vcf.py:25-350`) -
`VCF(config, handle)`, `write_header(contigs_lengths)`, `write_call(call) -> int`, `format_genotype`, `format_info`.

Pure text formatting of finished `SVCall`s: no kernel and no arithmetic beyond what the reference's writer does to a
call on its way out (END of precise deletions, SVLEN of sequence-resolved insertions, AC / SUPP_VEC over the samples,
the `minsvlen` cut on insertions, REF / ALT resolution against a reference handle, QUAL clamp).  It is here so that
parity can be stated on the VCF line itself - POS, SVLEN, SVTYPE, GT and every other column - and it reproduces the
reference's quirks because a drop-in must: with a reference handle the IUPAC clean-up table is applied to the whole
ALT column (`<INS>` -> `<INN>`, `<DEL>` -> `<NEL>`, `chrY` in a BND ALT -> `chrN`).
`reference_handle` is any object with pysam's `fetch(contig, start, end) -> str` (raising KeyError / ValueError for an
unknown contig / range).  The force-calling side (`--genotype-vcf`, vcf.py:352-481) is here too: `read_svs_iter` (targets
from a VCF), `rewrite_header_genotype`, `rewrite_genotype`.
"""
from __future__ import annotations

from collections import Counter

IUPAC_AMBIGUOUS = "RYSWKMBDHV"
_IUPAC_TO_N = str.maketrans(IUPAC_AMBIGUOUS, "N" * len(IUPAC_AMBIGUOUS))

# ##-lines after the contig table: (section, ID, Number, Type, Description); Number/Type None = not part of the line
_ALT = [("INS", "Insertion"), ("DEL", "Deletion"), ("DUP", "Duplication"), ("INV", "Inversion"),
        ("BND", "Breakend; Translocation")]
_FORMAT = [("GT", "1", "String", "Genotype"), ("GQ", "1", "Integer", "Genotype quality"),
           ("DR", "1", "Integer", "Number of reference reads"), ("DV", "1", "Integer", "Number of variant reads"),
           ("PS", "1", "Integer", "Phase-block, zero if none or not phased"),
           ("ID", "1", "String", "Individual sample SV ID for multi-sample output")]
_FILTER = [
    ("PASS", "All filters passed"), ("GT", "Genotype filter"), ("SUPPORT_MIN", "Minimum read support filter"),
    ("STDEV_POS", "SV Breakpoint standard deviation filter"), ("STDEV_LEN", "SV length standard deviation filter"),
    ("COV_MIN", "Minimum coverage filter"), ("COV_MIN_GT", "Minimum coverage filter (missing genotype)"),
    ("COV_CHANGE_DEL", "Coverage change filter for DEL"), ("COV_CHANGE_DUP", "Coverage change filter for DUP"),
    ("COV_CHANGE_INS", "Coverage change filter for INS"),
    ("COV_CHANGE_FRAC_US", "Coverage fractional change filter: upstream-start"),
    ("COV_CHANGE_FRAC_SC", "Coverage fractional change filter: start-center"),
    ("COV_CHANGE_FRAC_CE", "Coverage fractional change filter: center-end"),
    ("COV_CHANGE_FRAC_ED", "Coverage fractional change filter: end-downstream"),
    ("COV_VAR", "Coverage variance exceeded"), ("MOSAIC_VAF", "Mosaic variant allele fraction filter"),
    ("NOT_MOSAIC_VAF", "Variant allele fraction filter for non-mosaic"), ("ALN_NM", "Length adjusted mismatch filter"),
    ("STRAND_BND", "Strand support filter for BNDs"), ("STRAND", "Strand support filter for germline SVs"),
    ("STRAND_MOSAIC", "Strand support filter for mosaic SVs"), ("SVLEN_MIN", "SV length filter"),
    ("SVLEN_MIN_MOSAIC", "SV length filter for mosaic SVs (min)"), ("SVLEN_MAX_MOSAIC", "SV length filter for mosaic SVs (max)"),
    ("SINGLE_BREAK", "A single break point was detected but not classified as an SV."),
    ("INLINE_SA", "INLINE/CIGAR-based SV is mostly supported by SA reads"),
    ("MOSAIC_SV_CLOSE_EDGE", "For mosaic SVs, the location is close to the end of the read (either end)"),
    ("GT_FAILED", "Sniffles was unable to genotype this call.")]
_SVLENGTHS = ("SVLENGTHS", ".", "Integer", "Lengths of structural variation (all)")
_INFO_HEAD = [("PRECISE", "0", "Flag", "Structural variation with precise breakpoints"),
              ("IMPRECISE", "0", "Flag", "Structural variation with imprecise breakpoints"),
              ("MOSAIC", "0", "Flag", "Structural variation classified as putative mosaic"),
              ("SVLEN", "1", "Integer", "Length of structural variation")]
_INFO_TAIL = [
    ("SVTYPE", "1", "String", "Type of structural variation"), ("CHR2", "1", "String", "Mate chromsome for BND SVs"),
    ("SUPPORT", "1", "Integer", "Number of reads supporting the structural variation"),
    ("SUPPORT_INLINE", "1", "Integer", "Number of reads supporting an INS/DEL SV (non-split events only)"),
    ("SUPPORT_SA", "1", "Integer", "Number of reads supporting a DEL SV through supplementary alignments (split events)"),
    ("SUPPORT_LONG", "1", "Integer", "Number of soft-clipped reads putatively supporting the long insertion SV"),
    ("END", "1", "Integer", "End position of structural variation"),
    ("STDEV_POS", "1", "Float", "Standard deviation of structural variation start position"),
    ("STDEV_LEN", "1", "Float", "Standard deviation of structural variation length"),
    ("COVERAGE", ".", "Float", "Coverages near upstream, start, center, end, downstream of structural variation"),
    ("STRAND", "1", "String", "Strands of supporting reads for structural variant"),
    ("AC", ".", "Integer", "Allele count, summed up over all samples"),
    ("SUPP_VEC", "1", "String", "List of read support for all samples"),
    ("CONSENSUS_SUPPORT", "1", "Integer", "Number of reads that support the generated insertion (INS) consensus sequence"),
    ("RNAMES", ".", "String", "Names of supporting reads (if enabled with --output-rnames)"),
    ("VAF", "1", "Float", "Variant Allele Fraction"),
    ("COVERAGE_VAR", "1", "Float", "Variance of coverage across large events"),
    ("NM", ".", "Float", "Mean number of query alignment length adjusted mismatches of supporting reads"),
    ("PHASE", ".", "String", "Phasing information derived from supporting reads, represented as list of: "
                              "HAPLOTYPE,PHASESET,HAPLOTYPE_SUPPORT,PHASESET_SUPPORT,HAPLOTYPE_FILTER,PHASESET_FILTER"),
    ("LASM", "0", "Flag", "Local assembly used to detect the structural variant")]
_INFO_POPULATION = [("POPULATION_AF", "1", "Float", "Population Allele Frequency"),
                    ("POPULATION_SIZE", "1", "Integer", "Size of genotyped population for this variant")]


def _fast():
    from . import sv
    return sv._load_fast()


def _filters():
    from . import sv
    return sv.FILTERS


def format_info(key, value) -> str:
    """One INFO entry (vcf.py:26-37): floats with three decimals, lists joined, None as '.', True as a bare flag."""
    if isinstance(value, float):
        return f"{key}={value:.3f}"
    if isinstance(value, list):
        return f"{key}={','.join(value)}"
    if value is None:
        value = "."
    if value is True:
        return f"{key}"
    return f"{key}={value}"


def unpack_phase(phase) -> tuple:
    """(haplotype, phase set for the PS column) of a genotype's phase entry (vcf.py:40-51)."""
    if phase is None:
        hp, ps = None, "."
    else:
        try:
            hp, ps = phase
        except TypeError:
            hp, ps = phase, "."
    if ps is None or ps == "NULL":
        ps = "."
    return hp, ps


def format_genotype(gt, is_phased) -> str:
    """A sample column (vcf.py:54-83): 6-tuples come from single-sample calling, 7-tuples (with the per-sample SV id)
    from combine.  Phased notation only for 0/1 and 1/1 with a haplotype; haplotype "1" puts the ALT allele first."""
    a, b, qual, dr, dv, phase = gt[:6]
    hp, ps = unpack_phase(phase)
    sep = "/"
    if is_phased and hp is not None and (a, b) in ((0, 1), (1, 1)):
        sep = "|"
        if hp == "1":
            a, b = b, a
    cols = [f"{a}{sep}{b}", str(qual), str(dr), str(dv)]
    if is_phased:
        cols.append(str(ps))
    if len(gt) != 6:
        cols.append(str(gt[6]))
    return ":".join(cols)


class VCF:
    def __init__(self, config, handle):
        self.config = config
        self.handle = handle
        self.call_count = 0
        self.info_order = ["SVTYPE", "SVLEN", "END", "SUPPORT", "RNAMES", "COVERAGE", "STRAND"]
        if config.qc_nm_measure:
            self.info_order.append("NM")
        if config.dev_emit_sv_lengths:
            self.info_order.append("SVLENGTHS")
        self.default_genotype = config.genotype_none
        self.genotype_format = config.genotype_format
        if config.phase:
            self.genotype_format += ":PS"
        if config.mode == "combine":
            self.genotype_format += ":ID"
            self.default_genotype += ("NULL",)
        self.reference_handle = None
        self.header_str = ""

    # ---- header
    def write_raw(self, text, endl="\n"):
        self.handle.write(text)
        self.handle.write(endl)

    def write_header_line(self, text):
        self.write_raw("##" + text)

    def _typed(self, section, rows):
        for ident, number, typ, desc in rows:
            self.write_header_line(f'{section}=<ID={ident},Number={number},Type={typ},Description="{desc}">')

    def write_header(self, contigs_lengths):
        cfg = self.config
        self.write_header_line("fileformat=VCFv4.2")
        self.write_header_line(f"source={cfg.version}_{cfg.build}")
        self.write_header_line(f'command="{cfg.command}"')
        self.write_header_line(f'fileDate="{cfg.start_date}"')
        for contig, length in contigs_lengths:
            self.write_header_line(f"contig=<ID={contig},length={length}>")
        for ident, desc in _ALT:
            self.write_header_line(f'ALT=<ID={ident},Description="{desc}">')
        self._typed("FORMAT", _FORMAT)
        for ident, desc in _FILTER:
            self.write_header_line(f'FILTER=<ID={ident},Description="{desc}">')
        self._typed("INFO", _INFO_HEAD)
        if cfg.dev_emit_sv_lengths:
            self._typed("INFO", [_SVLENGTHS])
        self._typed("INFO", _INFO_TAIL)
        if cfg.combine_population:
            self._typed("INFO", _INFO_POPULATION)
        samples = "\t".join(sample_id for _, sample_id in cfg.sample_ids_vcf)
        self.write_raw(f"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{samples}")

    # ---- records
    def _sample_columns(self, call):
        """Per-sample genotype strings, allele count and support vector (vcf.py:229-245)."""
        cols, supp, ac = [], [], 0
        phased = self.config.phase
        for internal_id, _ in self.config.sample_ids_vcf:
            gt = call.genotypes.get(internal_id)
            if gt is None:
                cols.append(format_genotype(self.default_genotype, phased))
                supp.append("0")
                continue
            cols.append(format_genotype(gt, phased))
            if gt[0] != "." and gt[4] > 0:
                ac += gt[0] + gt[1]
                supp.append("1")
            else:
                supp.append("0")
        return cols, ac, "".join(supp)

    def _resolve_sequences(self, call) -> bool:
        """REF / ALT against the reference handle (vcf.py:302-342).  False: the call is not emitted (too many N)."""
        cfg = self.config
        ref = self.reference_handle
        if not cfg.symbolic and call.svtype == "DEL" and ref is not None and abs(call.svlen) <= cfg.max_del_seq_len:
            try:   # the deleted bases behind the last reference base before the SV
                call.ref = ref.fetch(call.contig, call.pos - 1, call.pos - call.svlen)
                call.alt = call.ref[0]
            except (KeyError, ValueError):
                call.ref, call.alt = "N", f"<{call.svtype}>"
            else:
                if "N" in call.ref and Counter(call.ref)["N"] / len(call.ref) > cfg.max_unknown_pct:
                    return False
        if cfg.symbolic:
            call.ref = "N"
            if call.svtype != "BND":
                call.alt = f"<{call.svtype}>"
            return True
        if ref is not None and call.ref == "N":
            start = max(0, call.pos - 1)
            try:
                call.ref = ref.fetch(call.contig, start, start + 1)
            except (KeyError, ValueError):
                pass
            else:
                if call.svtype == "INS" and call.alt != "<INS>":
                    call.alt = call.ref + call.alt
                elif call.svtype == "BND" and call.alt != "<BND>":
                    call.alt = (call.ref + call.alt[1:]) if call.alt.startswith("N") else call.alt[:-1] + call.ref
            call.ref = call.ref.translate(_IUPAC_TO_N)
            call.alt = call.alt.translate(_IUPAC_TO_N)   # also hits symbolic ALTs and BND mate names
        return True

    def write_call(self, call) -> int:
        """One record; returns 1 if a line was written.  Mutates the call like the reference's writer does."""
        cfg = self.config
        if call.is_single_break:
            return 0
        pos = call.pos if call.pos > 0 else 1
        end = pos + abs(call.svlen) if (call.precise and call.svtype == "DEL") else call.end

        sample_cols, ac, supp_vec = self._sample_columns(call)
        if len(cfg.sample_ids_vcf) > 1:
            call.set_info("AC", ac)
            call.set_info("SUPP_VEC", supp_vec)
            if int(supp_vec) == 0:
                return 0
            if ac == 0:
                call.filter = "GT"

        if call.svtype == "INS":
            if call.svlen != len(call.alt) and not cfg.symbolic and call.alt != "<INS>":
                call.svlen = len(call.alt)      # SVLEN follows the resolved sequence (before the REF base is prepended)
            if call.svlen < cfg.minsvlen:
                return 0

        bnd = call.svtype == "BND"
        core = {
            "SVTYPE": call.svtype,
            "SVLEN": None if bnd else call.svlen,
            "SVLENGTHS": None if bnd or not call.svlens else ",".join(map(str, call.svlens)),
            "END": None if bnd else end,
            "SUPPORT": call.support,
            "RNAMES": call.rnames if cfg.output_rnames else None,
            "COVERAGE": ",".join(str(c) for c in (call.coverage_upstream, call.coverage_start, call.coverage_center,
                                                  call.coverage_end, call.coverage_downstream)),
            "STRAND": ("+" if call.fwd > 0 else "") + ("-" if call.rev > 0 else ""),
            "NM": call.nm,
        }
        fields = ["PRECISE" if call.precise else "IMPRECISE"]
        vaf = call.get_info("VAF")
        if cfg.mosaic and (vaf if vaf is not None else 0) <= cfg.mosaic_af_max:
            fields.append("MOSAIC")
        fields.extend(format_info(k, core[k]) for k in self.info_order if core[k] is not None)
        fields.extend(format_info(k, call.info[k]) for k in sorted(call.info) if call.info[k] is not None)

        if not self._resolve_sequences(call):
            return 0
        if call.qual is not None:
            call.qual = max(0, min(60, call.qual))
        row = [call.contig, pos, cfg.id_prefix + call.id, call.ref, call.alt, "." if call.qual is None else call.qual,
               call.filter, ";".join(fields), self.genotype_format] + sample_cols
        self.write_raw("\t".join(str(v) for v in row))
        self.call_count += 1
        return 1

    # ---- merged records without the objects
    def can_write_merged(self) -> bool:
        """The group-table writer (`write_merged`) serves the plain multi-sample merge: two or more sample columns in the order of
        `config.snf_input_info`, no reference FASTA attached, no SVLENGTHS, no pair relabelling."""
        cfg = self.config
        ids = [i for i, _ in cfg.sample_ids_vcf]
        return (self.reference_handle is None and cfg.mode == "combine" and len(ids) > 1 and not cfg.dev_emit_sv_lengths
                and not getattr(cfg, "combine_pair_relabel", False) and not getattr(cfg, "combine_consensus", False)
                and ids == [s["internal_id"] for s in cfg.snf_input_info] and _fast() is not None)

    def merged_text_options(self) -> dict:
        cfg = self.config
        return dict(phase=bool(cfg.phase), symbolic=bool(cfg.symbolic), mosaic=bool(cfg.mosaic) and 0 <= cfg.mosaic_af_max,
                    output_rnames=bool(cfg.output_rnames), nm="NM" in self.info_order, minsvlen=int(cfg.minsvlen),
                    genotype_format=self.genotype_format)

    def write_merged(self, part, sort: bool = True) -> int:
        """One task's merged records as `candstore.execute_many(..., text_writer=self)` returned them - `(text, line_off, pos)` - in the
        order `sorted(calls, key=pos)` gives the objects (stable), or as they are.  Returns the number of lines written."""
        import numpy as np
        text, off, pos = part
        if len(pos) == 0:
            return 0
        if not sort or bool((pos[1:] >= pos[:-1]).all()):      # in order already (candstore formats them so): one slice
            n = int(np.count_nonzero(off[1:] > off[:-1]))
            raw = getattr(self.handle, "buffer", None)      # a text file over a binary one (open(path, "w")): the bytes go as they are
            if raw is not None and getattr(self.handle, "encoding", "").lower().replace("-", "") == "utf8":
                self.handle.flush()
                raw.write(memoryview(text)[int(off[0]):int(off[-1])])
            else:
                self.handle.write(str(memoryview(text)[int(off[0]):int(off[-1])], "utf-8"))
            self.call_count += n
            return n
        order = np.argsort(pos, kind="stable")
        n = 0
        out = []
        for k in order.tolist():
            a, b = int(off[k]), int(off[k + 1])
            if b > a:
                out.append(text[a:b])
                n += 1
        self.handle.write(b"".join(out).decode("utf-8"))
        self.call_count += n
        return n

    def can_write_records(self) -> bool:
        """The record-table writer serves the plain single-sample case: no reference FASTA attached (REF / ALT stay "N" /
        the consensus), one sample column, no SVLENGTHS."""
        cfg = self.config
        return (self.reference_handle is None and len(cfg.sample_ids_vcf) == 1 and cfg.sample_ids_vcf[0][0] == 0
                and not cfg.dev_emit_sv_lengths and cfg.mode != "combine" and _fast() is not None)

    def write_records(self, res, ti, order) -> int:
        """The records `order` (indices into `res.calls`, output order) of a finalized single-sample batch as VCF lines: the
        text `write_call` produces for the `SVCall` objects of the same records (sv.materialize_candidates + apply_final),
        formatted straight from the record table by the C extension.  Returns the number of lines written."""
        import numpy as np
        cfg = self.config
        opts = dict(id_prefix=cfg.id_prefix, mosaic=bool(cfg.mosaic), mosaic_af_max=float(cfg.mosaic_af_max),
                    output_rnames=bool(cfg.output_rnames), nm="NM" in self.info_order, phase=bool(cfg.phase), symbolic=bool(cfg.symbolic),
                    minsvlen=int(cfg.minsvlen), genotype_format=self.genotype_format,
                    genotype_none=format_genotype(self.default_genotype, cfg.phase))
        text, n = _fast().vcf_records(np.ascontiguousarray(res.calls), np.ascontiguousarray(order, np.int64),
                                      np.ascontiguousarray(res.rnames, np.uint32), np.ascontiguousarray(res.alt_pool, np.uint8),
                                      ti.qnames, ti.ps_names, ti.contig, int(ti.task_id), ti.contig_names, list(_filters()), opts)
        self.handle.write(text.decode("utf-8"))
        self.call_count += n
        return n

    # ---- force calling: targets in, the same lines with this sample's genotype out (vcf.py:352-481)
    def read_svs_iter(self):
        """Target SVs of `--genotype-vcf`: one `SVCall` per record of `self.handle` (text or bytes lines); header lines
        are collected in `self.header_str`.  Type and length come from REF / ALT unless INFO gives SVTYPE (`TRA` = BND),
        SVLEN, END; a BND's mate is parsed from its ALT.  `call.id` is the 1-based line number, like the reference."""
        from . import sv
        self.header_str = ""
        for line_index, line in enumerate(self.handle, 1):
            try:
                if isinstance(line, bytes):
                    line = line.decode("utf-8")
                text = line.strip()
                if text == "":
                    raise IndexError("string index out of range")     # the reference indexes the empty string here
                if text[0] == "#":
                    self.header_str += text + "\n"
                    continue
                chrom, pos, _, ref, alt, qual, flt, info_text = line.split("\t")[:8]
                info = {}
                for item in info_text.split(";"):
                    if "=" in item:
                        key, value = item.split("=")
                    else:
                        key, value = item, True
                    info[key] = value
                call = sv.SVCall(contig=chrom, pos=int(pos) - 1, id=line_index, ref=ref, alt=alt, qual=int(qual) if qual != "." else None,
                                 filter=flt, info=info, svtype=None, svlen=None, end=None, rnames=None, qc=True, postprocess=None,
                                 genotypes=None, precise=None, support=0, fwd=0, rev=0, nm=-1)
                if len(alt) > len(ref):
                    call.svtype, call.svlen, call.end = "INS", len(alt), call.pos
                else:
                    call.svtype, call.svlen = "DEL", -len(ref)
                    call.end = call.pos + call.svlen
                if "SVTYPE" in info:
                    call.svtype = "BND" if info["SVTYPE"] == "TRA" else info["SVTYPE"]
                if "SVLEN" in info:
                    call.svlen = int(info["SVLEN"])
                if "END" in info:
                    call.end = int(info["END"])
                if call.svtype == "BND":
                    parts = alt.replace("]", "[").split("[")
                    if len(parts) <= 2:
                        raise ValueError("BND ALT not formatted according to VCF 4.2 specifications")
                    mate_contig, mate_pos = parts[1].split(":")
                    call.bnd_info = sv.SVCallBNDInfo(mate_contig=mate_contig, mate_ref_start=int(mate_pos), is_first=(alt[0] == "N"),
                                                     is_reverse=("]" in alt))
                call.raw_vcf_line, call.raw_vcf_line_index = text, line_index
            except Exception as e:
                raise ValueError(f"Error parsing input VCF: Line {line_index}: {e}") from e     # the reference exits here
            yield call

    def rewrite_header_genotype(self, orig_header: str):
        lines = orig_header.split("\n")
        cfg = self.config
        lines[1:1] = [f"##genotypeSource={cfg.version}_{cfg.build}", f'##genotypeCommand="{cfg.command}"',
                      f'##genotypeFileDate="{cfg.start_date}"']
        for ident, number, typ, desc in _FORMAT[:4]:          # GT, GQ, DR, DV: added if the input does not declare them
            if not any(f"##FORMAT=<ID={ident}," in ln for ln in lines):
                lines.insert(len(lines) - 2, f'##FORMAT=<ID={ident},Number={number},Type={typ},Description="{desc}">')
        self.write_raw("\n".join(lines), endl="")

    def rewrite_genotype(self, svcall):
        """The target's own line (first eight columns) with FORMAT and this sample's genotype: the matched candidate's
        if there is one and it was genotyped, else the reference-allele genotype from the coverage (parallel.py:355-366)."""
        match = svcall.genotype_match_sv
        gt = match.genotypes[0] if (match is not None and len(match.genotypes) > 0) else svcall.genotypes[0]
        cols = svcall.raw_vcf_line.split("\t")[:8] + [self.config.genotype_format, format_genotype(gt, self.config.phase)]
        self.write_raw("\t".join(cols))
