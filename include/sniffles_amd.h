/*
 * sniffles_amd.h - C-ABI of the MI355X-native Sniffles2 hot path.
 *
 * The reference (fritzsedlazeck/Sniffles, pure Python) has no FFI; the seam this
 * library plugs into is a set of Python call sites.  Each entry
 * point below names the reference interface it replaces.  All pointers are plain
 * host pointers to caller-owned contiguous buffers (numpy arrays), borrowed for the
 * duration of the call; results are library-owned until snf_batch_destroy().
 *
 * A "batch" is a set of independent contig tasks (reference: one CallTask per
 * contig, src/sniffles/sniffles:298-358) executed together in fused launches on
 * one GPU.  A batch of one task is exactly one reference Task.
 *
 * Error convention: every function returns 0 on success, non-zero on failure;
 * snf_last_error() returns a thread-local message (reference convention: raise
 * -> ErrorResult, src/sniffles/parallel.py:750-752).
 */
#ifndef SNIFFLES_AMD_H
#define SNIFFLES_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNF_ABI_VERSION 5   /* 4: snf_batch_timing_every; 5: snf_batch_open */

#define SNF_SVLEN_NONE INT32_MIN /* Lead.svlen is None */
#define SNF_SEQ_NONE (-1)        /* Lead.seq is None   */
#define SNF_PS_NONE (-1)         /* Lead.phase_set is None */

/* sv.ALL_TYPES order, src/sniffles/sv.py:31-33 */
enum snf_svtype {
  SNF_INS = 0, SNF_DEL = 1, SNF_DUP = 2, SNF_INV = 3, SNF_BND = 4,
  SNF_SINGLE_LEFT = 5, SNF_SINGLE_RIGHT = 6, SNF_NTYPES = 7
};

/* Lead.source values, src/sniffles/leadprov.py:591-670, 247-281, 113-131 */
enum snf_source { SNF_SRC_INLINE = 0, SNF_SRC_SPLIT_PRIM = 1, SNF_SRC_SPLIT_SUP = 2, SNF_SRC_BND_SA = 3 };

/* SVCall.filter strings, src/sniffles/postprocessing.py:133-600, genotyping.py:138,176 */
enum snf_filter {
  SNF_F_PASS = 0, SNF_F_STDEV_POS, SNF_F_STDEV_LEN, SNF_F_SINGLE_BREAK, SNF_F_SVLEN_MIN,
  SNF_F_STRAND_BND, SNF_F_COV_CHANGE_DEL, SNF_F_COV_CHANGE_DUP, SNF_F_COV_CHANGE_INS,
  SNF_F_INLINE_SA, SNF_F_COV_VAR, SNF_F_COV_CHANGE_FRAC_US, SNF_F_COV_CHANGE_FRAC_SC,
  SNF_F_COV_CHANGE_FRAC_CE, SNF_F_COV_CHANGE_FRAC_ED, SNF_F_SUPPORT_MIN, SNF_F_GT_FAILED,
  SNF_F_GT, SNF_F_COV_MIN_GT, SNF_F_ALN_NM, SNF_F_MOSAIC_VAF, SNF_F_SVLEN_MAX_MOSAIC,
  SNF_F_STRAND, SNF_F_STRAND_MOSAIC, SNF_F_SVLEN_MIN_MOSAIC, SNF_F_COV_MIN,
  SNF_F_NOT_MOSAIC_VAF, SNF_F_MOSAIC_SV_CLOSE_EDGE, SNF_F_COUNT
};

/* per-task status: the reference's own failure modes are part of its behaviour */
enum snf_task_status {
  SNF_TASK_OK = 0,
  /* postprocessing.coverage raises UnboundLocalError('end') when the first candidate of a
     task is a BND (postprocessing.py:84-106): the task yields an ErrorResult */
  SNF_TASK_ERR_UNBOUND_END = 1
};

/*
 * Constants that parameterise the hot path (config.py:103-619).
 * Filled by the Python host from a SnifflesConfig-compatible namespace.
 */
typedef struct snf_config {
  /* clustering, config.py:250-262,417 */
  int32_t cluster_binsize;
  int32_t cluster_merge_pos;
  int32_t cluster_merge_bnd;
  int32_t cluster_resplit_binsize;
  double cluster_r;
  double cluster_repeat_h;
  double cluster_repeat_h_max;
  double cluster_merge_len;
  /* lengths / support, config.py:218-229,508-517,556-557 */
  int32_t minsvlen;
  int32_t minsvlen_screen;
  int32_t minsvlen_hard_cap;
  int32_t minsupport; /* -1 == "auto" */
  double minsupport_auto_base;
  double minsupport_auto_mult;
  double minsupport_auto_regional_coverage_weight;
  int32_t long_ins_length;
  int32_t long_del_length;
  int32_t long_dup_length;
  int32_t long_inv_length;
  double long_ins_rescale_base;
  double long_ins_rescale_mult;
  double long_del_coverage;
  double long_dup_coverage;
  int32_t dev_longer_del;
  int32_t dev_longer_dup;
  /* consensus, config.py:402,549-552 */
  int32_t consensus_max_reads_bin;
  int32_t consensus_min_reads;
  int32_t consensus_kmer_len;
  int32_t consensus_kmer_skip_base;
  double consensus_kmer_skip_seqlen_mult;
  /* calls / coverage, config.py:545-546,413,583 */
  int32_t precise;
  int32_t coverage_binsize;
  int32_t coverage_updown_bins;
  /* genotyping, config.py:270-271,569 */
  int32_t genotype_ploidy;
  int32_t genotype_min_z_score;
  double genotype_error;
  /* QC, config.py:224-228,443 */
  int32_t qc_stdev;
  int32_t qc_stdev_abs_max;
  int32_t qc_strand;
  int32_t qc_coverage;
  int32_t qc_bnd_filter_strand;
  int32_t qc_nm;
  int32_t qc_nm_measure;
  int32_t pass_only;
  double qc_coverage_max_change_frac;
  double qc_nm_mult;
  double dev_inline_sa_support_max;
  double dev_min_dup_vaf;
  /* mosaic, config.py:346-362 */
  int32_t mosaic;
  int32_t mosaic_min_reads;
  int32_t mosaic_use_strand_thresholds;
  int32_t max_svlen_mosaic;
  int32_t mosaic_qc_invdup_min_length;
  int32_t mosaic_qc_nm;
  int32_t mosaic_qc_strand;
  int32_t mosaic_include_germline;
  double mosaic_af_max;
  double mosaic_af_min;
  int32_t dev_min_close_edge_dist;
  int32_t dev_minreads_extra;
  int32_t dev_maxsvlen_extra;
  int32_t _pad0;
  double dev_min_read_close_edge_prop;
  /* switches */
  int32_t dev_min_leads_cluster;
  int32_t repeat;
  int32_t phase;
  int32_t detect_large_ins;
  int32_t no_consensus;
  int32_t symbolic;
  int32_t dev_no_resplit;
  int32_t dev_no_resplit_repeat;
  int32_t dev_output_candidates;
  int32_t mode_call_sample; /* config.mode == "call_sample" (parallel.py:204) */
  double phase_conflict_threshold;
  /* combine, config.py:301-310 */
  int32_t combine_match;
  int32_t combine_match_max;
  int32_t combine_separate_intra;
  int32_t _pad1;
  double combine_pctseq;
  /* CallTask.execute: which calls leave the task and in which order (parallel.py:265-271; read by SNF_OUT_EXECUTE) */
  int32_t no_qc;
  int32_t sort;
} snf_config_t;

/*
 * One contig task, struct-of-arrays, rows in arrival (BAM) order
 * (Lead dataclass, src/sniffles/leadprov.py:34-56; LeadProvider, :358-472).
 */
typedef struct snf_task_input {
  int32_t task_id;      /* Task.id, used in SVCall ids (sv.py:563) */
  int32_t sv_id_start;  /* Task.sv_id on entry */
  int32_t contig_len;   /* len(lead_provider.coverage) */
  int32_t ps_null_rank; /* rank of the literal phase-set string "NULL", -1 if absent */
  double qc_nm_threshold; /* config.qc_nm_threshold side channel, leadprov.py:577-578 */

  int64_t n_leads;
  const int32_t* ref_start;
  const int32_t* ref_end;
  const int32_t* qry_start;
  const int32_t* qry_end;
  const int32_t* svlen; /* SNF_SVLEN_NONE == None */
  const int32_t* read_len;
  const uint32_t* qname_id; /* interned read_qname (equality only) */
  const uint32_t* read_id;
  const int32_t* ps_rank;        /* order-preserving rank of phase_set string, SNF_PS_NONE == None */
  const int32_t* mate_contig;    /* order-preserving rank of bnd_info.mate_contig */
  const int32_t* mate_ref_start;
  const int32_t* seq_len; /* SNF_SEQ_NONE == None */
  const int64_t* seq_off; /* into seq_pool */
  const double* nm;
  const uint8_t* svtype;
  const uint8_t* strand; /* 0 '+', 1 '-' */
  const uint8_t* mapq;
  const uint8_t* source;
  const uint8_t* hap; /* int(Lead.hap) in 0..2 */
  const uint8_t* is_sa;
  const uint8_t* bnd_is_first;
  const uint8_t* bnd_is_reverse;

  int64_t seq_pool_len;
  const uint8_t* seq_pool;

  /* alignment records: coverage (leadprov.py:510) and REF hap counts (leadprov.py:387-398) */
  int64_t n_reads;
  const int32_t* read_start;
  const int32_t* read_end;
  const uint8_t* read_hp;

  /* tandem repeats, padded, sorted (util.py:121-147); n_tr < 0 == None */
  int64_t n_tr;
  const int32_t* tr_start;
  const int32_t* tr_end;

  /* LeadProvider._mask_N_coverage (leadprov.py:420-443): with --reference the coverage vector is set to 0 wherever the
   * reference base is 'N' (byte 78).  The mask as intervals [nmask_start[k], nmask_end[k]) - sorted, disjoint, inside
   * [0, contig_len); n_nmask == 0: no mask.  Reading the FASTA is the host's business (sniffles_amd.soa.nmask_intervals
   * turns a sequence into the intervals). */
  int64_t n_nmask;
  const int32_t* nmask_start;
  const int32_t* nmask_end;
} snf_task_input_t;

/*
 * One SVCall (src/sniffles/sv.py:87-223), integer/float fields only; strings
 * (id, filter, alt for symbolic/BND, PHASE, read names) are formatted by the host.
 */
typedef struct snf_call {
  int32_t task_index; /* index into the batch's task list */
  int32_t sv_id;      /* id = f"{svtype}.{sv_id:X}S{task_id:X}" */
  int32_t svtype;
  int32_t pos;
  int32_t end;
  int32_t svlen;
  int32_t support;
  int32_t support_long; /* INFO SUPPORT_LONG (INS), -1 absent */
  int32_t support_sa;   /* INFO SUPPORT_SA (DEL), -1 absent */
  int32_t qual;
  int32_t precise;
  int32_t fwd;
  int32_t rev;
  int32_t qc;
  int32_t filter; /* enum snf_filter */
  int32_t cov[5]; /* upstream, start, center, end, downstream */
  int32_t sa_count; /* Cluster.sa_counts[0] */
  int32_t n_leads;  /* len(cluster.leads) at call time */
  double sa_frac;   /* Cluster.sa_counts[1] */
  double nm;
  double stdev_pos;
  double stdev_len; /* NaN == None (BND) */
  /* BND (SVCallBNDInfo, sv.py:36-43) */
  int32_t mate_contig;
  int32_t mate_ref_start;
  int32_t bnd_is_first;
  int32_t bnd_is_reverse;
  /* genotype tuple (a, b, GQ, DR, DV, (hp, ps)), genotyping.py:182 */
  int32_t gt_set; /* 0: genotypes dict empty (GT_FAILED) */
  int32_t gt_a;
  int32_t gt_b;
  int32_t gt_gq;
  int32_t gt_dr;
  int32_t gt_dv;
  int32_t gt_hp; /* -1 None, else 1/2 */
  int32_t gt_ps; /* -1 None, -2 literal "NULL", else ps rank */
  double vaf;    /* INFO VAF, NaN absent */
  /* INFO PHASE = f"{hp},{ps},{hp_support},{ps_support},{hp_filter},{ps_filter}" */
  int32_t ph_set;
  int32_t ph_hp;
  int32_t ph_ps; /* -2 "NULL", else rank */
  int32_t ph_hp_support;
  int32_t ph_ps_support;
  int32_t ph_hp_pass;
  int32_t ph_ps_pass;
  /* ALT: alt_len >= 0 -> bytes in the alt pool; -1 -> "<SVTYPE>" / BND string from fields */
  int32_t alt_len;
  int64_t alt_off;
  /* supporting read names (qname ids, ascending), rnames = list(support_set), sv.py:555 */
  int64_t rn_off;
  int32_t rn_len;
  /* provenance of the cluster (Cluster.start/end/seed, cluster.py:27-39) */
  int32_t cluster_start;
  int32_t cluster_end;
  /* index of the seed bin among ALL occupied bins of its (task, svtype) sequence (the last field of Cluster.id,
     cluster.py:261; the reference emits it only in --dev-dump-clusters / trace output).  -1 = not provided: batches with
     dev_min_leads_cluster >= 2 drop the leads of single-lead bins in front of the sort (SNF_NO_PREFILTER=1 turns that
     off); snf_batch_fetch_clusters always returns it */
  int32_t cluster_seed_index;
} snf_call_t;

typedef struct snf_result {
  int64_t n_calls;
  const snf_call_t* calls; /* candidate order: task, svtype, seed, resplit */
  int64_t alt_pool_len;
  const uint8_t* alt_pool;
  int64_t rnames_len;
  const uint32_t* rnames;
  int64_t n_tasks;
  const int32_t* task_status;             /* enum snf_task_status per task */
  const int64_t* task_call_off;           /* n_tasks+1 offsets into calls */
  const double* coverage_average_total;   /* Task.coverage_average_total per task */
} snf_result_t;

typedef struct snf_batch snf_batch_t;

/* library / device ---------------------------------------------------------------- */
int snf_abi_version(void);
const char* snf_last_error(void);
/* number of visible HIP devices (0 => every compute entry point fails loudly) */
int snf_device_count(void);

/* The library keeps a few things of finished batches for the next one (HBM slabs, pinned result buffers, HIP streams: creating
 * them costs more than a contig's kernels).  They are released on their own when an allocation fails; this hands them back
 * explicitly - `device` < 0: every device.  Returns the bytes released. */
int64_t snf_trim_caches(int device);

/* batch lifecycle -------------------------------------------------------------------
 * replaces: LeadProvider.__init__/record_lead/record_hap_ref/build_leadtab
 * (src/sniffles/leadprov.py:361-472) as the container of one task's signatures. */
int snf_batch_create(const snf_config_t* cfg, int device, snf_batch_t** out);
/* registers one task; may be called n_tasks times.  The task's arrays are BORROWED until snf_batch_upload returns
 * (nothing is copied here). */
int snf_batch_add_task(snf_batch_t* b, const snf_task_input_t* task);
/* Device-resident hand-over from the extraction (declared below: snf_extract_*): the leads, sequence pool and read table a
 * finished snf_extract_run left in HBM become a task of the batch without a round trip through the host.  `meta` supplies
 * what the extraction does not know: task_id, sv_id_start, contig_len and the tandem repeats (host arrays, borrowed until
 * snf_batch_upload like those of snf_batch_add_task); its other pointers are ignored.  The extraction handle must stay
 * alive and must not be re-run until snf_batch_upload has returned; it has to live on the batch's device. */
struct snf_extract;
int snf_batch_add_task_device(snf_batch_t* b, struct snf_extract* x, const snf_task_input_t* meta);
/* host -> HBM: host threads validate the tasks and stage their columns into one pinned (process-wide, grow-only) arena,
 * two large copies move it, kernels derive the packed per-lead records; after this the caller's buffers are no longer
 * referenced.  A task the reference could not have produced (svtype / hap codes, sequence ranges, a read outside its
 * region, the byte '-' in a sequence) fails the call. */
int snf_batch_upload(snf_batch_t* b);
/* snf_batch_create + snf_batch_add_task x n_tasks + snf_batch_upload and, with `run`, the first device work of the batch, in ONE
 * call: run = SNF_RUN_NONE, SNF_RUN_CANDIDATES (snf_batch_call_candidates), or SNF_RUN_PASS | an output mode
 * (snf_batch_set_output(mode) + snf_batch_pass).  What a worker loop calls for task k + 1 from a helper thread while it turns task
 * k's records into objects (reference: the worker processes of `sniffles:495-530` get that overlap from being several): a caller
 * whose host language serialises its threads between foreign calls (CPython's GIL) then pays one hand-over per task, not five.
 * On failure nothing is left behind (*out = NULL).  The tasks' arrays are borrowed until the call returns. */
#define SNF_RUN_NONE 0
#define SNF_RUN_CANDIDATES 1
#define SNF_RUN_PASS 0x100   /* | SNF_OUT_CANDIDATES or SNF_OUT_EXECUTE (| SNF_OUT_DEVICE) */
int snf_batch_open(const snf_config_t* cfg, int device, const snf_task_input_t* tasks, int32_t n_tasks, int run, snf_batch_t** out);
void snf_batch_destroy(snf_batch_t* b);

/* replaces Task.call_candidates (src/sniffles/parallel.py:104-127):
 * cluster.resolve + merge_inner/resplit/resplit_bnd (cluster.py:85-353),
 * Cluster.get_sa_count, sv.call_from/resolve_bnd (sv.py:497-639),
 * postprocessing.coverage (postprocessing.py:69-130). Asynchronous on the batch stream. */
int snf_batch_call_candidates(snf_batch_t* b);

/* replaces Task.finalize_candidates (src/sniffles/parallel.py:129-201):
 * qc_sv, qc_sv_support, annotate_sv (phase_sv, genotype_sv, INS consensus
 * consensus.novel_from_reads), qc_sv_post_annotate, rescue_phasing. */
int snf_batch_finalize(snf_batch_t* b);

/* snf_batch_call_candidates + snf_batch_finalize back to back - the two statements of CallTask.execute
 * (reference src/sniffles/parallel.py:264-266) - enqueued as ONE unit.  Same results as the two calls.  From the second pass of a
 * handle on (same input: every launch size is known) the pass is replayed from a HIP graph captured per result configuration
 * (output mode, result memory): one host call per pass instead of ~40 launches on four streams.  SNF_NO_GRAPH=1 keeps it eager.
 * Returns 0, or 1 with snf_last_error(). */
int snf_batch_pass(snf_batch_t* batch);

/* What a stage-1 fetch returns (set before snf_batch_finalize; default SNF_OUT_CANDIDATES):
 *   SNF_OUT_CANDIDATES  every candidate of every task that did not raise, in candidate order - the list
 *                       Task.finalize_candidates returns (src/sniffles/parallel.py:129-201; its early exits are commented out)
 *   SNF_OUT_EXECUTE     what CallTask.execute keeps of it (parallel.py:265-271): the calls with `qc` set (all of them under
 *                       config.no_qc), per task stably sorted by pos when config.sort - the statement
 *                       `svcalls = [s for s in svcalls if s.qc]` / `sorted(svcalls, key=pos)` on the device, so that only the
 *                       records, ALT bytes and read names the parent process will see cross PCIe
 *   | SNF_OUT_DEVICE    the result stays in HBM until the fetch (instead of being stored straight into pinned host memory by
 *                       the kernels): required for snf_batch_export_device
 * snf_call_t.alt_off / rn_off of the returned records index the returned pools. */
enum snf_output { SNF_OUT_CANDIDATES = 0, SNF_OUT_EXECUTE = 1, SNF_OUT_DEVICE = 2 };
int snf_batch_set_output(snf_batch_t* b, int mode);

/* Where the stage-1 result lands on the host: `block` receives [ records | read names ] and `alt` the ALT section, both
 * written by the kernels themselves (zero-copy), so the memory must be page-locked for the device: the library registers the
 * two ranges (hipHostRegister) the first time it sees them and unpins them at snf_batch_destroy - switching between a few
 * segments (one per pass in flight) costs nothing after the first round.  Meant for memory
 * ANOTHER PROCESS maps as well - one process per GPU, the parent (the reference's `Main`, parallel.py:757 receives whole
 * results) reads every worker's result from a shared-memory segment without a copy and without funnelling it through one
 * GPU's PCIe link.  A result that does not fit fails the fetch with the sizes needed.  NULL, 0, NULL, 0: the library's own
 * pinned buffers again.  Call it between passes (it waits for the batch's streams).  The memory must stay mapped until
 * snf_batch_destroy (the registration is only dropped there); a range given again with a larger size is registered anew. */
int snf_batch_set_result_memory(snf_batch_t* b, void* block, int64_t block_bytes, void* alt, int64_t alt_bytes);

/* device -> host of the call records (blocks until the batch's streams are idle).
 * stage 0: the candidates after call_candidates (no ALT); 1: after finalize, as selected by snf_batch_set_output.
 * snf_batch_call_candidates + snf_batch_finalize enqueue without waiting for the device; this is the one host wait of a pass.
 * Pointers valid until the next call on the batch or snf_batch_destroy. */
int snf_batch_fetch(snf_batch_t* b, int stage, snf_result_t* out);
/* number of candidate calls of the last snf_batch_call_candidates (all tasks; what Task.sv_id advances by, sv.py:563), as of
 * the last fetch / sync; -1 on error */
int64_t snf_batch_n_candidates(snf_batch_t* b);

/* final gather across GPUs (the parent receives whole results, parallel.py:757): the finalized result of the
 * batch as ONE block of bytes in HBM, copied device-to-device into `dst_device` (e.g. a torch CUDA tensor handed to RCCL):
 *   [ snf_call_t records | read names (uint32) at layout->off_rnames | ALT bytes at layout->off_alt ]   (layout->bytes in all)
 * Needs SNF_OUT_DEVICE.  Fails when cap_bytes is too small (layout is filled in regardless, so the caller can grow).
 * Blocks until the copy is done. */
typedef struct snf_export_layout {
  int64_t n_calls, rnames_len, alt_pool_len;
  int64_t off_rnames, off_alt, bytes;
} snf_export_layout_t;
int snf_batch_export_device(snf_batch_t* b, void* dst_device, int64_t cap_bytes, snf_export_layout_t* layout);

/* replaces SNFile.annotate_block_coverages (src/sniffles/snf.py:249-267): the coverage vector
 * (leadprov.py:451,510) zero-padded to a multiple of `binsize`, averaged per bin and rounded with
 * Python's round().  out[i] is the value of bin first_bin + i (positions [j*binsize, (j+1)*binsize)),
 * or -1 when the bin lies beyond the padded vector (the reference's IndexError: the bin is skipped).
 * Formed from the task's sparse read table on the device (the read index is built by snf_batch_upload). */
int snf_batch_block_coverage(snf_batch_t* b, int32_t task_index, int32_t binsize, int64_t first_bin,
                             int64_t n_bins, int32_t* out);

/* replaces postprocessing.coverage(calls, lead_provider) (src/sniffles/postprocessing.py:69-130) for calls that are
 * not the batch's own candidates: the target SVs of GenotypeTask.execute (src/sniffles/parallel.py:353).
 * svtype: SNF_* codes (anything but INS / BND takes end = pos + |svlen|); cov: 5 ints per call, in/out
 * (upstream, start, center, end, downstream) - a sample outside the coverage vector keeps its input value (the
 * reference ignores the IndexError).  *status = 1 when a BND comes before any other call (UnboundLocalError in
 * the reference: the calls from there on are left untouched).  *coverage_mean = coverage.mean() of the task.
 * Needs snf_batch_call_candidates first. */
int snf_batch_coverage_calls(snf_batch_t* b, int32_t task_index, int64_t n, const int32_t* svtype, const int32_t* pos,
                             const int32_t* svlen, const uint8_t* bnd_is_first, int32_t* cov, int32_t* status,
                             double* coverage_mean);

/* replaces postprocessing.genotype_sv(svcall, config) (src/sniffles/postprocessing.py:607-623 -> genotyping.py:62-241) for
 * calls that are not a batch's own: the candidates of SNF files older than 2.5.3 that CombineTask.execute genotypes again
 * (--reqc, parallel.py:507-508).  In / out on the records: reads svtype, svlen, support, support_sa, cov[5], filter, qc and
 * the current phase (gt_hp, gt_ps, ph_*); writes gt_set, gt_a, gt_b, gt_gq, gt_dr, gt_dv, gt_hp, gt_ps, vaf, filter, qc,
 * ph_hp_pass (a call without usable coverage gets filter GT_FAILED and keeps its genotype, as in the reference). */
int snf_genotype_batch(const snf_config_t* cfg, int device, snf_call_t* calls, int64_t n);

/* Seam B3 / debugging: the clusters of the candidate stage as the reference's cluster.resolve
 * (src/sniffles/cluster.py:219-353) sees them, copied to the host after snf_batch_call_candidates.
 *   stage 0: seed clusters (one per 100-bp bin that passes dev_min_leads_cluster, cluster.py:238-275)
 *   stage 1: after the adaptive merge scan - the list the reference dumps with --dev-dump-clusters (cluster.py:278-324)
 *   stage 2: what resolve() yields - after merge_inner / resplit / resplit_bnd (cluster.py:326-353)
 * Clusters come in the reference's order (task, svtype, seed; resplit order inside a merged cluster); the leads of
 * cluster i are lead[lead_off[i] .. lead_off[i+1]) in the reference's list order: rows of the task's input table
 * (stage 2: the row of the head of a fused lead) with the svlen they carry at that stage (fused by merge_inner at stage 2).
 * Pointers are valid until the next call on the batch. */
typedef struct snf_clusters {
  int64_t n_clusters;
  const int32_t* task_index;
  const int32_t* svtype;
  const int32_t* start;        /* Cluster.start / end / seed (cluster.py:262-266, 302) */
  const int32_t* end;
  const int32_t* seed;
  const int32_t* seed_index;   /* index of the seed bin among the sorted bins of its (task, svtype): Cluster.id */
  const int32_t* n_leads_long; /* len(cluster.leads_long) (INS), 0 otherwise */
  const uint8_t* repeat;
  const int64_t* lead_off;     /* n_clusters + 1 */
  int64_t n_leads;
  const int32_t* lead;         /* row in the task's input arrays */
  const int32_t* lead_svlen;   /* SNF_SVLEN_NONE never occurs here (leads_long are counted, not listed) */
} snf_clusters_t;
int snf_batch_fetch_clusters(snf_batch_t* b, int stage, snf_clusters_t* out);

/* per-kernel timing (HIP events on the kernels' own streams, recorded around the launches of the
 * last call_candidates+finalize pass). names[i] points to a static string.
 * The brackets are recorded on every n-th pass of a handle (default 8, the first pass included; snf_batch_timing_every(b, 1):
 * every pass, 0: none): two event records per launch keep a stream from issuing its launches back to back, which costs a
 * whole-genome pass 3-6 %.  The one kernel bench.py states the roofline on (the LARGE consensus class, the last kernel
 * of its stream) is bracketed on every pass.  A pass that was not sampled reports only that kernel. */
int snf_batch_timing_every(snf_batch_t* b, int n);
int snf_batch_timing_count(snf_batch_t* b);
int snf_batch_timing_get(snf_batch_t* b, int i, const char** name, float* ms, int64_t* algo_bytes);
/* the same as MEANS over all passes since the last reset (what bench.py states the roofline with: a kernel's average launch
 * duration over the timed region); `passes` = passes that contributed */
int snf_batch_timing_mean_reset(snf_batch_t* b);
int snf_batch_timing_mean_count(snf_batch_t* b);
int snf_batch_timing_mean_get(snf_batch_t* b, int i, const char** name, float* ms, int64_t* algo_bytes, int* passes);
/* block until everything queued on the batch stream has finished */
int snf_batch_sync(snf_batch_t* b);

/* replaces edlib.align(a, b)["editDistance"] as used by SVGroup.align_call
 * (src/sniffles/sv.py:280-289) and snfp (src/sniffles/snfp.py:103): global unit-cost
 * edit distance for n_pairs string pairs. a_off/b_off have n_pairs+1 entries. */
int snf_edit_distance_batch(int device, const uint8_t* a_pool, const int64_t* a_off,
                            const uint8_t* b_pool, const int64_t* b_off,
                            int64_t n_pairs, int32_t* out_dist);
/* the same with edlib's `k` argument (edlib.align(a, b, k=...)): max_dist[i] >= 0 bounds the distances of interest of pair
 * i - the alignment is banded accordingly (Ukkonen) and out_dist[i] is -1 when the distance exceeds it, as edlib reports;
 * max_dist[i] < 0 (or max_dist == NULL): exact.  Both entry points stage through a persistent per-device arena (HBM +
 * pinned mirror + stream): no allocation once it has reached its working size. */
int snf_edit_distance_batch_k(int device, const uint8_t* a_pool, const int64_t* a_off,
                              const uint8_t* b_pool, const int64_t* b_off,
                              int64_t n_pairs, const int32_t* max_dist, int32_t* out_dist);

/*
 * Multi-sample combine: group assignment of cluster.resolve_block_groups (src/sniffles/cluster.py:356-390)
 * including SVGroup.align_call (src/sniffles/sv.py:280-289).  One problem = one call of resolve_block_groups
 * (one svtype, one flush window of CombineTask.execute, src/sniffles/parallel.py:524-534); problems are
 * independent and run concurrently.  Candidates are given in the order of `svcands`; the stable sort by
 * support (descending), the greedy nearest-group search with running means (SVGroup.add_candidate,
 * sv.py:297-318) and the on-demand edit distance happen on the GPU.  out_group[i] = index of candidate i's
 * group in the list `groups_initial + [new groups in creation order]`.  SVGroup.call (sv.py:320-481) is
 * host-side bookkeeping over the resulting membership (sniffles_amd/sv.py).
 */
typedef struct snf_combine_problem {
  int32_t svtype;
  int32_t n_cands;
  int32_t n_groups;     /* len(groups_initial) */
  int32_t n_sample_ids; /* sample_internal_id values are in [0, n_sample_ids) */
  const int32_t* pos;
  const int32_t* svlen;
  const int32_t* support;
  const int32_t* sample_id;
  const int32_t* mate_contig;    /* BND: order-free id of bnd_info.mate_contig (equality only) */
  const int32_t* mate_ref_start; /* BND */
  const int64_t* alt_off;        /* n_cands + 1 offsets into alt_pool: SVCall.alt */
  const uint8_t* alt_pool;
  /* groups_initial: SVGroup state (sv.py:226-241) */
  const double* g_pos_mean;
  const double* g_len_mean;
  const double* g_mate_mean;     /* bnd_mate_ref_start_mean */
  const int32_t* g_size;         /* len(group.candidates) */
  const int32_t* g_mate_contig;
  const int64_t* g_alt_off;      /* n_groups + 1: group.candidates[0].alt */
  const uint8_t* g_alt_pool;
  const int64_t* g_samples_off;  /* n_groups + 1: group.included_samples */
  const int32_t* g_samples;
  int32_t* out_group;            /* n_cands */
  /* Optional chain of flush windows (CombineTask.execute, parallel.py:516-558): the candidates are the concatenation
   * of n_windows windows, window w = candidates [win_off[w], win_off[w+1]).  After a window every group with
   * abs(pos_mean - win_bin[w]) < win_thr[w]  (win_thr = max(size * 0.5, combine_overlap_abs)) stays active for the
   * next window ("keep"), the others are flushed; out_group then numbers the groups over the whole chain
   * (groups_initial first, new groups in creation order).  n_windows == 0: one window, nothing is flushed. */
  int32_t n_windows;
  const int32_t* win_off;        /* n_windows + 1 */
  const int32_t* win_bin;        /* curr_bin of the flush */
  const double* win_thr;
} snf_combine_problem_t;

int snf_combine_resolve_batch(const snf_config_t* cfg, int device, const snf_combine_problem_t* problems,
                              int64_t n_problems);
/* measurement: kernel time (HIP events) of the last snf_combine_resolve_batch on `device` and what it aligned -
 * stats4[0] alignments (align_call evaluations), [1] bytes of the aligned strings, [2] cells of their full DP matrices,
 * [3] bytes staged host -> HBM */
int snf_combine_last_stats(int device, double* kernel_ms, int64_t* stats4);

/*
 * Multi-sample combine, second half: SVGroup.call (src/sniffles/sv.py:320-481) and the keep / flush bookkeeping of
 * CombineTask.execute (src/sniffles/parallel.py:536-572) for ALL groups of a merge at once, over a columnar candidate
 * table - no per-object work.  The caller gives every candidate as one numeric record, the membership
 * snf_combine_resolve_batch produced as lists in the order SVGroup.add_candidate saw the candidates (window by window,
 * inside a window support descending, stable), and the flush windows.  Per group the device replays the running
 * pos_mean (the same three roundings per candidate as sv.py:297-318), walks the windows the group is alive in
 * (abs(pos_mean - win_bin[w]) < win_thr[w] keeps it), and evaluates the call: confidence rules, medians
 * (util.median), means (util.mean_or_none_round: Python round() = rint), the ALT choice of insertions, exact
 * statistics.stdev, and which candidate's genotype represents its sample.  Strings (ids, ALT, read names, phase tuples)
 * stay on the host and are looked up through the indices returned here.
 */
#define SNF_NONE_I32 INT32_MIN
typedef struct snf_group_cand {
  int32_t pos, svlen, end, support;
  int32_t qual;            /* int(c.qual); SNF_NONE_I32: None */
  int32_t fwd, rev;
  int32_t cov[5];          /* coverage_upstream, _start, _center, _end, _downstream; SNF_NONE_I32: None */
  int32_t gq, dr, dv;      /* genotypes[0]; without one: 0, 0, support (sv.py:392) */
  int32_t sample;          /* sample_internal_id */
  int32_t alt_len;         /* len(c.alt) */
  int8_t gt_a, gt_b;       /* -1: "." */
  uint8_t qc, pass, precise, is_ins;   /* c.qc, c.filter == "PASS", c.precise, svtype == "INS" */
  uint8_t _pad[2];
} snf_group_cand_t;          /* 76 bytes */

typedef struct snf_group_call_config {
  int32_t n_samples;               /* len(config.snf_input_info) */
  int32_t no_qc;
  int32_t combine_low_confidence_abs;
  int32_t combine_output_filtered;
  int32_t dev_combine_medians;
  int32_t minsvlen_screen;
  double combine_high_confidence;
  double combine_low_confidence;
} snf_group_call_config_t;

typedef struct snf_group_out {
  int32_t flush_win;       /* window whose keep test the group failed; -1: still active at the end of its chain */
  int32_t emit;            /* 1: SVGroup.call returns a call, 0: None, -1: a member lies behind the group's flush (caller error) */
  int32_t n_pass, n_present;
  int32_t pos, svlen, end; /* of the combined call */
  int32_t alt_member;      /* position in `member` of the candidate whose ALT the call takes */
  int32_t qual;            /* SNF_NONE_I32: None */
  int32_t support, fwd, rev;
  int32_t cov[5];          /* SNF_NONE_I32: None */
  int32_t precise;
  int32_t n;               /* candidates; 1: util.stdev returns the int 0 */
  double stdev_pos, stdev_len;
} snf_group_out_t;           /* 96 bytes */

/* group g = member[group_off[g] .. group_off[g+1]) (indices into cand / cand_win, add order); cand_win[c] = window of
 * candidate c, non-decreasing along a group's members; group_win_hi[g] = one past the last window of g's chain;
 * member_chosen[k] = 1 iff member k carries the genotype of its sample (sv.py:395-404);
 * member_pos_mean[k] = group.pos_mean after member k was added.  Returns 0, or 1 (invalid arguments, no device, or a member
 * that lies behind the flush of its group). */
int snf_combine_call_groups(const snf_group_call_config_t* cfg, int device, int64_t n_groups, const int64_t* group_off,
                            const int32_t* member, int64_t n_cands, const snf_group_cand_t* cand, const int32_t* cand_win,
                            const int32_t* group_win_hi, int64_t n_windows, const int32_t* win_bin, const double* win_thr,
                            snf_group_out_t* out, uint8_t* member_chosen, double* member_pos_mean);

/* ---------------------------------------------------------------------------------------------------------------------
 * Seam B4: consensus.novel_from_reads(best_lead, other_leads, klen, skip, skip_repetitive)
 * (reference src/sniffles/consensus.py:280-394, called from postprocessing.py:63) for a batch of independent
 * problems.  Only the `.seq` of the leads is read, so a problem is the best sequence plus the other sequences, all
 * given as (offset, length) into one byte pool.  `skip[p]` is the sampling step of the other reads, `skip_repetitive[p]` the
 * one of the best read's anchors (NULL: the same as `skip`, which is what the reference's call site passes,
 * postprocessing.py:60-61).  The output of problem p has the length of its best sequence and is written to
 * out_pool + out_off[p]  (out_off[p+1] - out_off[p] == best_len[p]).
 * klen 1..8.  Problems within klen <= 7, best_len < 65000, <= 512 other sequences, <= 500 sampled positions (best_len / skip),
 * skip_repetitive == skip and a pool without the byte '-' (the reference's gap symbol: a read base '-' IS a gap there,
 * consensus.py:317-380) run on the workgroup kernels; every other problem is served by the literal thread kernels (anchor table,
 * one thread per other read, one thread per column - the reference's rows with '-' as the gap byte).
 * Returns 0, or 1 with snf_last_error().
 */
int snf_consensus_batch(int device, int klen, const uint8_t* seq_pool, int64_t seq_pool_len, int64_t n_problems,
                        const int64_t* best_off, const int32_t* best_len, const int32_t* skip, const int32_t* skip_repetitive,
                        const int64_t* others_index, /* n_problems + 1: range of a problem in others_off / others_len */
                        const int64_t* others_off, const int32_t* others_len,
                        uint8_t* out_pool, const int64_t* out_off);

/* ------------------------------------------------------------------------------------------------------------------
 * Signature extraction: inflated BAM alignment records of one contig -> the task input above.
 * Replaces, for one `Region(contig, start, end)`, LeadProvider.build_leadtab (src/sniffles/leadprov.py:445-472) and
 * everything it drives per alignment: iter_region (:474-578: read filter, read ids, `coverage[s:e] += 1`,
 * record_hap_ref, the NM side channel config.qc_nm_threshold), read_iterindels (:580-655), Lead.for_bnd (:57-132),
 * CIGAR_analyze (:144-178), read_itersplits (:221-355) and sv.classify_splits (src/sniffles/sv.py:649-782).
 * The pysam layer (bam.fetch, AlignedSegment properties, get_tag) is replaced by parsing the raw records on the GPU:
 * one wave per alignment record, 64 CIGAR operations per step.
 *
 * The caller (host) inflates BGZF, finds the record boundaries and interns the strings the hot path only compares:
 * read names (rank per record) and contig names (hash table -> rank), both ranks in Python str order.
 */
typedef struct snf_extract_config {
  int32_t mapq;                    /* config.mapq (20) */
  int32_t min_alignment_length;    /* config.min_alignment_length (1000) */
  int32_t exclude_flags;           /* config.exclude_flags, -1 == None */
  int32_t minsvlen_screen;         /* int(minsvlen_screen_ratio * minsvlen), config.py:517 */
  int32_t long_ins_length;         /* 2500 */
  int32_t dev_seq_cache_maxlen;    /* 50000 */
  int32_t max_splits_base;         /* 3 */
  int32_t detect_large_ins;        /* bool */
  int32_t advanced_tags;           /* qc_nm_measure or phase (leadprov.py:477) */
  int32_t dev_keep_lowqual_splits; /* bool */
  double max_splits_kb;            /* 0.1 */
} snf_extract_config_t;

typedef struct snf_extract_input {
  const uint8_t* records;   /* inflated BAM alignment records back to back, each starting with its block_size field */
  int64_t records_len;
  const int64_t* rec_off;   /* n_records + 1 byte offsets into `records` */
  int64_t n_records;        /* file order; records of other contigs / unmapped records are skipped */
  const uint32_t* qname_rank; /* per record: order-preserving rank of the read name */
  int32_t region_ref_id;    /* BAM refID of the region's contig */
  int32_t region_rank;      /* rank of the region's contig name (same ranking as contig_rank) */
  int32_t region_start;     /* Region.start, Region.end (0-based, half open) */
  int32_t region_end;
  uint32_t read_id_offset;  /* LeadProvider.read_id on entry, modulo 2^32: read ids are only compared for equality inside a
                               task, so the reference's task.id * 10^k offsets (which exceed 32 bits) are reduced by the host */
  int32_t n_contigs;        /* header contigs: 64-bit FNV-1a of the name, ascending, and the rank of that name */
  const uint64_t* contig_hash;
  const int32_t* contig_rank;
} snf_extract_input_t;

typedef struct snf_extract_result {
  snf_task_input_t task;    /* leads (record_lead order), seq pool, reads, ps_null_rank, qc_nm_threshold are filled;
                               task_id / sv_id_start / contig_len / tandem repeats are the caller's */
  int64_t n_ps;             /* phase sets: ps_value[rank] is the PS tag whose str() has that rank;                */
  const int64_t* ps_value;  /* the entry at task.ps_null_rank stands for the literal "NULL" (tag absent)          */
  uint32_t read_id;         /* LeadProvider.read_id on exit */
  int64_t read_count;
  float ms_count;           /* kernel time of the counting pass / the emitting pass (HIP events) */
  float ms_emit;
  int64_t algo_bytes;       /* record header + name + CIGAR + aux bytes of the region's records + sequence bytes out */
} snf_extract_result_t;

typedef struct snf_extract snf_extract_t;
int snf_extract_create(const snf_extract_config_t* cfg, int device, snf_extract_t** out);
int snf_extract_upload(snf_extract_t* x, const snf_extract_input_t* in); /* host -> HBM */
int snf_extract_run(snf_extract_t* x);   /* a record the reference would raise on fails the call (ErrorResult) */
int snf_extract_result(snf_extract_t* x, snf_extract_result_t* out); /* library-owned until destroy / next run; the first
                                                                        call after a run copies the columns to the host */
/* counts, phase-set table, NM threshold and timings only - no column is copied to the host */
int snf_extract_result_meta(snf_extract_t* x, snf_extract_result_t* out);
/* the same columns as DEVICE pointers (out->... point into the HBM of *device), no copy: what snf_batch_add_task_device uses */
int snf_extract_device_view(snf_extract_t* x, snf_task_input_t* out, int* device);
void snf_extract_destroy(snf_extract_t* x);
const char* snf_extract_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SNIFFLES_AMD_H */
