"""Hot-path configuration with the reference's attribute names and defaults.

Mirrors the parts of `SnifflesConfig` (reference `src/sniffles/config.py:103-619`)
that parameterise clustering, calling, QC, genotyping and consensus (SURVEY.md
Appendix B).  The CLI/argparse surface is out of scope; the
constructor takes the same option names as keyword arguments, e.g.
`SnifflesConfig(mosaic=True, minsupport="auto", minsvlen="~50")`, and applies the
same derivations as `SnifflesConfig.__init__` (`config.py:507-617`).

A real reference `SnifflesConfig` instance can be passed anywhere this class is
accepted: only attribute access is used (`sniffles_amd.abi.config_struct`).
"""
from __future__ import annotations

_DEFAULTS = dict(
    # main / filter args (config.py:165-247)
    phase=True, minsupport="3", minsupport_auto_mult=None, minsvlen="~50", minsvlen_screen_ratio=0.9,
    no_qc=False, pass_only=False, qc_stdev=True, qc_stdev_abs_max=500, qc_strand=False, qc_coverage=1,
    long_ins_length=2500, long_del_length=50000, long_inv_length=10000, long_del_coverage=0.66,
    long_dup_length=50000, long_dup_coverage=1.33, qc_bnd_filter_strand=True,
    phase_conflict_threshold=0.1, detect_large_ins=True,
    # cluster args (config.py:253-262)
    cluster_binsize=100, cluster_r=2.5, cluster_repeat_h=1.5, cluster_repeat_h_max=1000.0,
    cluster_merge_pos=150, cluster_merge_len=0.22, cluster_merge_bnd=1000,
    # genotype args (config.py:267-274)
    genotype_ploidy=2, genotype_error=0.05,
    # combine args (config.py:296-316)
    combine_match=250, combine_match_max=1000, combine_separate_intra=False, combine_pctseq=0.7,
    combine_high_confidence=0.0, combine_low_confidence=0.2, combine_low_confidence_abs=2,
    combine_null_min_coverage=5, combine_output_filtered=False, combine_support_threshold=3,
    combine_pair_relabel=False, combine_pair_relabel_threshold=20, combine_consensus=False, combine_population=None,
    dev_combine_medians=False, combine_close_handles=False,
    # SNF container (config.py:272, 327, 477-479, 527)
    sample_id=None, output_rnames=False, snf=None,
    # VCF writer (config.py:166-170, 242, 332)
    vcf=None, reference=None, max_del_seq_len=50000, max_unknown_pct=0.5,
    # contig selection of the main program (config.py:168-176, util.py:147-162)
    all_contigs=False, contig=None, threads=4, regions_by_contig=None,
    # read filter of the extraction (config.py:190-215, 533-536); None: derived below
    mapq=None, min_alignment_length=None, exclude_flags=None, max_splits_kb=0.1, max_splits_base=3, dev_keep_lowqual_splits=False,
    # postprocess args (config.py:325-334)
    no_consensus=False, symbolic=False,
    # mosaic args (config.py:343-362)
    mosaic=False, mosaic_af_max=0.218, mosaic_af_min=0.05, mosaic_qc_invdup_min_length=500,
    mosaic_qc_nm=True, mosaic_qc_nm_mult=1.66, mosaic_qc_coverage_max_change_frac=-1.0, mosaic_qc_strand=True,
    mosaic_include_germline=False, max_svlen_mosaic=50000, mosaic_min_reads=3, mosaic_use_strand_thresholds=10,
    # developer args (config.py:388-446)
    consensus_max_reads_bin=10, dev_no_resplit=False, dev_no_resplit_repeat=False, repeat=False, qc_nm=False,
    qc_nm_mult=1.66, qc_coverage_max_change_frac=-1.0, coverage_updown_bins=5, cluster_binsize_combine_mult=5,
    cluster_resplit_binsize=20, dev_no_qc=False, dev_filter=False, dev_output_candidates=None,
    dev_min_leads_cluster=-1, dev_min_dup_vaf=1 / 6.0, dev_longer_del=200000, dev_longer_dup=200000,
    dev_minreads_extra=5, dev_maxsvlen_extra=10000, dev_inline_sa_support_max=0.80,
    dev_min_close_edge_dist=500, dev_min_read_close_edge_prop=0.75, dev_seq_cache_maxlen=50000,
    dev_emit_sv_lengths=False, dev_trace_read=False, dev_locasm_do=False, dev_dump_clusters=False,
)


class SnifflesConfig:
    GLOBAL = None

    def __init__(self, **kw):
        unknown = set(kw) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"unknown hot-path options: {sorted(unknown)}")
        for k, v in _DEFAULTS.items():
            setattr(self, k, kw.get(k, v))
        # derivations, config.py:507-617
        ms = str(self.minsvlen)
        if ms.startswith("~"):
            self.minsvlen_hard_cap = False
            self.minsvlen = int(ms[1:])
        else:
            self.minsvlen_hard_cap = True
            self.minsvlen = int(ms)
        self.minsvlen_screen = int(self.minsvlen_screen_ratio * self.minsvlen)
        if self.minsupport != "auto":
            self.minsupport = int(self.minsupport)
        if self.dev_no_qc:
            self.no_qc = True
        # --dev-no-qc also switches the read filter off (config.py:533-536)
        if self.mapq is None:
            self.mapq = 0 if self.dev_no_qc else 20
        if self.min_alignment_length is None:
            self.min_alignment_length = 0 if self.dev_no_qc else 1000
        if self.regions_by_contig is None:
            self.regions_by_contig = {}
        self.minsupport_auto_base = 1.5
        self.minsupport_auto_regional_coverage_weight = 0.75
        if self.minsupport_auto_mult is None:
            self.minsupport_auto_mult = 0.1
        self.coverage_binsize = self.cluster_binsize
        self.coverage_binsize_combine = self.cluster_binsize * self.cluster_binsize_combine_mult
        self.consensus_min_reads = 4
        self.consensus_kmer_len = 6
        self.consensus_kmer_skip_base = 3
        self.consensus_kmer_skip_seqlen_mult = 1.0 / 500.0
        self.long_ins_rescale_base = 1.66
        self.long_ins_rescale_mult = 0.33
        self.dev_longer_dup = min(self.long_dup_length * 4, self.dev_longer_dup)
        self.dev_longer_del = min(self.long_del_length * 4, self.dev_longer_del)
        self.genotype_min_z_score = 5
        if self.genotype_ploidy != 2:
            raise ValueError("Currently only genotype_ploidy 2 is supported")
        self.snf_block_size = 10 ** 5
        self.snf_format_version = "S2_rc4"       # config.py:31 - the format this package reads and writes
        self.version, self.build = "Sniffles2", "2.8.1-dev"   # the reference build whose behaviour is reproduced
        self.reqc = "auto"
        import datetime
        import sys
        self.start_date = datetime.datetime.now().strftime("%Y/%m/%d %H:%M:%S")   # config.py:472, 480
        self.command = " ".join(sys.argv)
        self.genotype_format = "GT:GQ:DR:DV"                                        # config.py:566-568
        self.genotype_none = (".", ".", 0, 0, 0, (None, None))
        self.genotype_null = (0, 0, 0, 0, 0, (None, None))
        self.sample_ids_vcf = [(0, "SAMPLE")]   # single-sample default (sniffles:177-181 uses the input's base name)
        self.sort = True
        self.combine_overlap_abs = 2500
        self.combine_min_size = 100
        self.precise = 25
        self.tandem_repeat_region_pad = 500
        self.id_prefix = "Sniffles2."
        self.phase_identifiers = ["1", "2"]
        if self.mosaic_include_germline:
            self.mosaic = True
        self.qc_nm_measure = self.qc_nm
        if self.mosaic:
            self.qc_nm_measure = self.qc_nm_measure or self.mosaic_qc_nm
            if self.cluster_merge_len == 0.22:
                self.cluster_merge_len = 0.27
        if self.dev_min_leads_cluster == -1:
            self.dev_min_leads_cluster = 1 if self.no_qc else 2
        self.mode = "call_sample"
        self.snf_input_info = []   # combine: [{'internal_id': i, ...}] per input sample (sniffles:371-420)
        # per-task side channel written by iter_region (leadprov.py:577-578)
        self.average_regional_nm = 0.02
        self.qc_nm_threshold = 0.02
        SnifflesConfig.GLOBAL = self
