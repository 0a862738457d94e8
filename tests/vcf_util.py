"""Shared by the VCF golden generator (reference writer) and the tests (this package's writer)."""

FIXED = dict(command="sniffles --input sample.bam --vcf out.vcf", start_date="2026/01/01 00:00:00")


class FakeFasta:
    """pysam.FastaFile stand-in: a deterministic pseudo-random reference with a sprinkling of N and IUPAC codes.
    fetch(contig, start, end) raises KeyError for an unknown contig and ValueError for a negative start (pysam)."""
    ALPHABET = "ACGTACGTACGTACGTACGTACGTACGTNRYK"

    def __init__(self, contigs):
        self.contigs = dict(contigs)

    def fetch(self, contig, start, end):
        if contig not in self.contigs:
            raise KeyError(contig)
        if start < 0 or end < start:
            raise ValueError("invalid coordinates")
        end = min(end, self.contigs[contig])
        return "".join(self.ALPHABET[((i * 2654435761) >> 9) & 31] for i in range(start, end))


# name -> (extra reference command line, config overrides set on both sides, with a reference handle)
VARIANTS = {
    "plain": ((), {}, False),
    "fasta": ((), {}, True),
    "symbolic_fasta": (("--symbolic",), {"symbolic": True}, True),
    "nophase": ((), {"phase": False}, False),
}
CASES = ["phase_rescue", "long_ins", "merge_inner", "bnd_stale_end", "consensus_quirks", "fuzz_4_2", "chr21_30x_mosaic",
         "single_leads_noqc", "chr20_30x_ont", "chr18_20x_auto_nm", "chr17_15x_noqc"]


def split_text(text):
    lines = text.split("\n")
    return [ln for ln in lines if ln.startswith("#")], [ln for ln in lines if ln and not ln.startswith("#")]


def key_columns(line):
    """(CHROM, POS, SVTYPE, SVLEN, GT columns) of a record - what BASELINE.json's north_star names."""
    f = line.split("\t")
    info = dict(kv.split("=", 1) for kv in f[7].split(";") if "=" in kv)
    return f[0], int(f[1]), info.get("SVTYPE"), info.get("SVLEN"), [c.split(":")[0] for c in f[9:]]
