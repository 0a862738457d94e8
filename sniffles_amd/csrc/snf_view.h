// snf_view.h - the HBM-resident state of one batch, passed by value to every kernel.
//
// Layout: everything is struct-of-arrays, concatenated over the tasks of the
// batch.  "lead" = one SV signature (reference Lead, leadprov.py:34-56).
//   N   leads (input order = arrival/BAM order inside each task)
//   R   alignment records (coverage + REF haplotype counts)
//   NF  leads that survive into seed clusters (normal, svlen != None)   <= N
//   NLL INS leads with svlen None that belong to a seed bin ("leads_long") <= N
#pragma once
#include "../../include/sniffles_amd.h"
#include "snf_rt.h"

namespace snf {

// device-side counters written by the pipeline (read back once per fetch)
struct Counts {
  int64_t n_valid, n_bins, n_seeds, NF, NLL, n_runs, n_clusters, n_rc, n_calls;
  int64_t n_ins_calls, alt_total, n_cons, tab_total, aln_total, n_cons_reads, rn_total;
  int64_t n_dirty_groups;
  int64_t n_kept;            // leads the occupancy prefilter lets through to the sort (a0_*; == NS whenever the prefilter is on)
  int64_t n_big[3];          // items the wave kernels handed to x_big<0 / 1 / 2> in this pass (sums of View::big_cnt, formed by z1_results)
  int64_t n_occ;             // window front end (snf_stage_window.h): occupied windows of this pass
  int64_t max_win;           // ... and the largest window (only formed at upload, w0_stats)
  int64_t n_big64;           // ... and the windows of more than 64 leads (upload): at most that many blocks need the large instance of w4s_segment
  unsigned long long n_w4big;       // blocks the small instance of w4s_segment handed on in this pass (View::w4_list)
  int64_t n_cons_fallback;   // consensus calls that do not fit the LDS workgroup kernel
  unsigned long long cons_bytes[4]; // algorithmic bytes of the ALT stage per class (0 fallback, 1 small, 2 large, 3 copy)
  unsigned long long n_cls[8];      // ALT work lists: 0 verbatim copy, 1 SMALL, 2-5 LARGE by work (2 = heaviest), 6 thread-kernel fallback,
                                    // 7 ROWS (aligned rows in HBM: beyond the LDS vote counters, or handed over by SMALL / LARGE at run time)
  unsigned long long pool_extra_used;
#ifdef SNF_CONS_PROFILE
  unsigned long long dbg[32];       // instrumented build only: cycles per phase of the consensus kernels (tools/cons_profile.sh)
#endif
#ifdef SNF_C1_PROFILE
  unsigned long long c1p[16];       // instrumented build only: phases of c1_mergeruns (100-MHz ticks summed over the waves' first lanes)
#endif
  int32_t overflow;  // scratch overflow flags
  int32_t alt_in_pinned;     // this pass's ALT bytes go straight to the pinned buffer (View::alt_pin) instead of the HBM pool: decided once the total is known (e3)
};

// genotype lookup entry for (normalised support, normalised coverage), built on the host with the
// same libm CPython uses (genotyping.py:124-183)
struct GtEntry {
  int8_t order0;  // index (0: 0/0, 1: 0/1, 2: 1/1) of the most likely genotype
  int8_t gq;
  int8_t z;
  int8_t _pad;
};
#define SNF_GT_N 251

// one seed-cluster lead, packed in sorted (L) order by a6_scatter: the wave kernels read these coalesced instead
// of gathering 8-10 scattered input columns through L[] -> input index
struct LeadRec {
  int32_t ref_start, ref_end, qry_start, qry_end;
  int32_t svlen, seq_len;          // seq_len < 0: Lead.seq is None (incl. the 10-per-bin cap)
  int64_t seq_off;
  uint32_t qname, read_id;
  int32_t ps, mate_pos;
  int32_t mate_contig, read_len;
  uint32_t orig;                   // input index
  // the eight small fields share one word, so that a record is exactly one 64-byte line
  uint32_t strand : 1, mapq : 8, source : 2, hap : 2, is_sa : 1, first : 1, rev : 1, svtype : 3;
};
static_assert(sizeof(LeadRec) == 64, "one cache line per packed lead record");

// one ALT work item, written by e3_conslist: everything a consensus / copy workgroup needs to start, in one record
// (instead of the chain cons_call -> callx -> F_seq_len/F_seq_off -> FI -> ...; the stage is latency-bound)
struct ConsDesc {
  int64_t best_off;   // pool offset of the best read
  int64_t alt_off;    // output offset in alt_pool
  int64_t aln_off;    // rows of this call in v.aln (L bytes per other read)
  int64_t read_off;   // first entry of this call in crl_off / crl_len / aln_kept
  int32_t L, n_others;
  int32_t skip;       // k-mer sampling step of this call (consensus_kmer_skip_base + int(L * mult), postprocessing.py:60)
  int32_t cls;
};

// merged cluster header, written by c4_clusters: what the refine kernels need to start on cluster c in one record
struct ClusterHdr { int32_t h, lo, n, grp; int32_t repeat, _pad[3]; };

// tile-sum slots of the fused flag -> scan -> emit chains (snf_fused.h)
enum { TS_BINS = 0, TS_SEEDS = 1, TS_LEADS = 2 /* and 3 */, TS_RUNS = 4, TS_CLUSTERS = 5, TS_REFINED = 6, TS_CALLS = 7, TS_RNAMES = 8, TS_KEEP = 9,
       TS_OUT = 10 /* 11, 12 */, TS_WIN = 13 /* 14 */, TS_WINC = 15 /* 16, 17 */, TS_SLOTS = 18 };

// totals and layout of the output block (f* kernels, snf_stage_out.h); copied to the pinned result block by z1_results
struct OutHdr {
  int64_t n_out, rn_out;              // records, read names (uint32)
  int64_t off_rn, bytes;              // sections of the block: [records | read names], 256-byte aligned; the ALT bytes are a section
                                      // of their own (all candidates, candidate order: written by the ALT kernels themselves)
  int32_t in_pinned;                  // 1: the kernels stored the block straight into pinned host memory
  int32_t _pad;
};

struct CallX {  // per-call internals that are not part of snf_call_t
  int32_t rc;       // refined cluster id
  int32_t cluster;  // merged cluster id
  int32_t flo, fn;  // range of the refined cluster in the F arrays
  int32_t best;     // F position of the consensus best lead, -1 none
  int32_t n_others;
  int32_t do_cons;
  int32_t cons_id;
  int64_t alt_off;
  int32_t rn_nq;    // distinct read names d2 left sorted in w1[flo..] (the rest come from leads_long): input of d3_rnames_emit
  // aggregates over the call's leads that QC / phasing need (LeadAgg, snf_stage_final.h), formed by d2w_call while the leads
  // are in its registers - e1w_finalize used to gather every lead's 64-byte record a second time for them (193 MB per pass)
  int32_t ag_valid; // 1: set (wave path, <= 64 leads); 0: whoever finalizes the call collects them itself
  int32_t ag_nstrands, ag_close_edge, ag_hp_val, ag_hp_support, ag_hp_other, ag_ps_val, ag_ps_support, ag_ps_other, ag_has_nm;
  int32_t _pad;
  double ag_nm_mean;
};

// Sums of compute_metrics over ALL leads of a cluster (kept while the cluster has fewer than 200 leads - the reference then samples
// every lead - and the sums fit 63 bits): sum of svlen, sum / sum of squares of ref_start - x0, x0 = ref_start of the cluster's first
// lead, hi = end of its lead range.  s2 == ~0: not kept, a merge reads the leads.
struct ClusterSums { int64_t sum, s1; uint64_t s2; int32_t x0, hi; };

struct View {
  snf_config_t cfg;
  int32_t T;          // tasks
  int32_t run_gap;    // merge-scan run cut (bp); <0: whole group serial
  int32_t wave_path;  // 1: gfx950 wave-cooperative kernels own the small clusters (thread kernels skip them)
  int32_t cons_thread_only;  // 1: a sequence of the batch holds the byte '-' (the reference's gap symbol): every consensus call takes
                             //    the literal thread kernels e4 / e5 / e6, which keep the reference's rows (consensus.py:317-380)
  int32_t _pad_cto;
  // stand-alone consensus seam (snf_consensus_batch): sampling step of the other reads / of the best read's anchors per
  // consensus id, instead of the pipeline's formula (postprocessing.py:60-61 passes the same value for both); null in a batch
  const int32_t* cons_skip_arr; const int32_t* cons_skiprep_arr;
  int32_t prof;       // SNF_PROF=1: phase cycle counters in e45w_consensus
  int32_t merge_reread;   // SNF_MERGE_REREAD=1 (tests): the merge walk reads the leads of every merged cluster instead of adding the kept sums
  int64_t N, R, NTR;
  int64_t NS;         // positions that go through the sort and the stages behind it: N, or (prefilter) the leads of (svtype, bin) cells with >= 2 leads
  int64_t pool_len, pool_cap;
  int64_t pool_extra_base;   // fused sequences reserved through Counts::pool_extra_used start here (pool_len, or behind the private slices)
  int64_t pool_slice;        // d1w_refine: bytes of fused-sequence space every resident wave owns at pool_len + blockIdx.x * pool_slice (0: none)
  Counts* cnt;

  // ---- tasks [T] / [T+1]
  const int32_t* t_task_id; const int32_t* t_sv_id_start; const int32_t* t_contig_len; const int32_t* t_ps_null;
  const double* t_qc_nm_thr;
  const int64_t* t_lead_off; const int64_t* t_read_off; const int64_t* t_tr_off;
  const int32_t* t_has_tr;
  int32_t* t_status; int64_t* t_call_off; double* t_cov_avg; int32_t* t_stale_end;
  unsigned long long* t_cov_sum;

  // ---- input leads [N]
  const int32_t *in_ref_start, *in_ref_end, *in_qry_start, *in_qry_end, *in_svlen, *in_read_len;
  const uint32_t *in_qname, *in_read_id;
  const int32_t *in_ps, *in_mate_contig, *in_mate_pos, *in_seq_len;
  const int64_t* in_seq_off;  // rebased into the batch pool
  const double* in_nm;
  const uint8_t *in_svtype, *in_strand, *in_mapq, *in_source, *in_hap, *in_is_sa, *in_first, *in_rev;
  const int32_t* lead_task;
  uint8_t* pool;  // [pool_cap]: input sequences, then fused sequences (merge_inner)

  // ---- reads [R] (starts sorted per task on input; ends sorted on device)
  const int32_t* r_start; const int32_t* r_end; const uint8_t* r_hp; const int32_t* r_task;
  uint64_t *rk_in, *rk_out; uint32_t *rv_in, *rv_out;  // end-sort scratch
  int32_t* re_sorted;        // [R] ends, ascending per task
  int32_t *rs_top, *re_top;  // every 256th entry of r_start / re_sorted (top level of the rank queries)
  int32_t *rs_mid, *re_mid;  // every 16th entry (the 16-ary descent of the pass's coverage queries, snf_exact.h::rank_upper_16ary)
  uint64_t* pc_s2;           // [R+1] prefix counts in start order: #(hp == 1) << 32 | #(hp == 2)
  uint64_t* pc_e2;           // [R+1] same in end order
  uint32_t* rflag;           // [R+1] scan scratch

  // ---- reference 'N' mask of the coverage vector (leadprov.py:420-443) and tasks whose coverage means need the exact walk
  const int32_t *nm_start, *nm_end;   // [NNM] mask intervals, concatenated over the tasks
  const int64_t* t_nm_off;            // [T+1] (null: no task has a mask)
  const int32_t* t_cov_exact;         // [T] 1: mask present or depth >= 65536 somewhere (uint16 wrap): means by cov_range_sum (snf_cov.h)

  // ---- tandem repeats [NTR]
  const int32_t* tr_start; const int32_t* tr_end; const int32_t* tr_pmax;

  // ---- occupancy prefilter (a0_*): a lead alone in its (task, svtype, bin) cell can never be part of a seed when
  // dev_min_leads_cluster >= 2 (cluster.py:262) - 80 % of the leads of a 30x genome - and is dropped in front of the sort.
  // Two bits per cell ("seen", "seen twice"); the words a pass touched are cleared again by the pass itself.
  int32_t prefilter;          // 1: on
  int32_t pf_spread;          // 1: neighbouring cells are spread over the L2 channels (pf_slot); SNF_PF_SPREAD=0 keeps them adjacent
  uint32_t* pf_bm;            // [pf_words] 16 cells per word
  const int64_t* t_cell_off;  // [T+1] first cell of task t (cells of a task: SNF_NTYPES x (contig_len / binsize + 1))
  uint64_t* pf_key;           // [N] sort key per input lead (uint32_t when key32), written by a1_keys
  uint32_t *pf_keep, *pf_scan;  // [N+1] keep flag per input lead / its exclusive scan
  // ---- window front end (snf_stage_window.h): leads bucketed by WINDOW = 2^win_bits consecutive bins of one (task, svtype), every
  // occupied window ordered and binned by one wave.  Replaces the prefilter + sort + a2..a7 chain when `front` is set.
  int32_t front;              // 1: on
  int32_t win_bits;           // W
  int64_t NW;                 // window slots of the batch (dense: per task SNF_NTYPES x windows of its contig)
  const int64_t* t_win_off;   // [T+1] first window of task t
  uint32_t *wcnt, *wfill;     // [NW+1] leads per window / fill cursor of the scatter (both zero between passes)
  uint32_t* wbase;            // [NW+1] bucket offsets (exclusive scan of wcnt)
  uint32_t* wlist;            // [4 x n_occ] occupied windows, ascending: {window, leads, bucket offset, task}
  uint32_t* blk_k0;           // [N / 64 + 2] per 64 positions of the bucket array: the first occupied window starting there or behind
  uint32_t *ws_seeds, *ws_nf, *ws_nl;   // [N / 64 + 2] per wave of w4s_segment: seeds / `leads` / `leads_long` it contributes, then their exclusive scans
  uint64_t* whead;            // [N] per seed head (bucket position): leads with a length | leads << 16 | hap 1 << 32 | hap 2 << 48
  uint64_t* whead2;           // [N] ... seed start | group << 32 | offset in the other lead list of the wave << 52
  // ---- stage A: binning (sorted position p in [0,NS))
  uint64_t *key_in, *key_out; uint32_t *val_in, *val_out;   // uint32_t keys when key32
  int key32, key_bin_bits, key_nbits;  // sort key = grp << key_bin_bits | bin; bit key_nbits set: lead outside its contig
  uint32_t *headflag, *headscan;  // [N+1] bin heads in sorted order / exclusive scan (bin ids)
  uint32_t *eligflag, *eligscan;  // [N+1] per bin: seeds a cluster / exclusive scan (seed ids)
  uint32_t *fN, *pN;              // [N+1] per sorted lead: goes to a seed's `leads` / scatter position
  uint32_t *fL, *pL;              // [N+1] per sorted lead: goes to a seed's `leads_long` / scatter position
  uint32_t *runflag, *runscan;    // [N+1] per seed: starts a merge-scan run / run ids
  uint32_t *clflag, *clscan;      // [N+1] per seed: head of a merged cluster / cluster ids
  uint32_t *rcflag, *rcscan;      // [N+1] per F slot: refined cluster starts here / rc ids
  uint32_t *cdflag, *cdscan;      // [N+1] per rc: produced a candidate call / call ids
  uint8_t* seqnull;          // [N] record_lead dropped the sequence (leadprov.py:406-408)
  int32_t* bin_lo;           // [N+1]
  uint64_t* bin_key;         // [N]
  uint16_t* bin_hap;         // [3N]
  uint8_t* bin_elig;         // [N]
  int32_t* grp_first_bin;    // [8T]
  uint32_t* L;               // [N] seed-cluster leads, (task, svtype, bin, arrival) order -> input index
  uint32_t* LL;              // [N] leads_long, same order
  LeadRec* Lrec;             // [N] packed records of L[] (same index)
  const LeadRec* in_rec;     // [N] the input columns interleaved per lead at upload (input order): one 64-B gather in a6
  ClusterHdr* chdr;          // [n_clusters]
  // ---- output stage (snf_stage_out.h): the calls a stage-1 fetch returns, compacted into one block
  int32_t out_mode;          // enum snf_output
  int32_t out_valid;         // host: the output stage of this pass has been enqueued (z1_results publishes its offsets)
  int32_t rn_defer;          // 1: the candidate stage only sizes the supporting read names; finalize writes them for the calls the
                             //    output keeps (SNF_OUT_EXECUTE: 70 % of the candidates' names would never be looked at); 2: late pass over ALL calls
  int32_t rn_from_src;       // this launch of f4w_emit writes the supporting read names straight from the leads (names deferred, kept calls only)
  uint32_t* o_scan;          // [n_calls+1] exclusive scan of the keep flags (defined for every call)
  int32_t *o_src, *o_dst, *o_key;   // [n_out] compacted index -> call index / final record index / pos (sort key)
  int64_t* o_rn;             // [n_out] offset inside the read-name section
  OutHdr* out_hdr;           // device
  OutHdr* res_out;           // pinned copy (z1_results)
  uint8_t* out_dev; int64_t out_dev_cap;   // block in HBM
  uint8_t* out_pin; int64_t out_pin_cap;   // block in pinned host memory (0: none)
  // result block in pinned host memory, written by z1_results at the end of each stage (no D2H copies to wait for)
  Counts* res_cnt; int32_t* res_status; int64_t* res_off; double* res_cov;
  int64_t* res_rn_total;     // pinned: total supporting-read-name count, written by d3_rnames (side stream)
  uint32_t *rnf, *rnp;       // [N+1] rn_len per call and its exclusive scan (own buffers: runs next to the ALT chain)

  // ---- stage B/C: seeds [n_seeds <= N]
  int32_t *seed_bin, *seed_lo, *seed_hi, *seedL_lo, *seedL_hi, *seed_start, *seed_grp;
  double *s_mean0, *s_stdev0; uint8_t* s_repeat0;     // seed metrics (kept for the serial fallback)
  double *c_mean, *c_stdev; uint8_t* c_repeat;        // live cluster metrics at head seeds
  int32_t *c_last, *c_end, *nxt, *prv;
  ClusterSums* c_ms;         // [N] at head seeds: the sums behind c_mean / c_stdev, so that a merge adds them instead of reading the leads again
  int32_t *run_first;        // [N+1]
  int32_t *run_last_head;    // [N]
  double *run_b_stdev, *run_b_absmean; uint8_t* run_b_repeat;
  int32_t* grp_dirty;        // [8T]
  int32_t *grp_seed_lo, *grp_seed_hi;  // [8T]

  // ---- stage D: merged clusters / refined clusters / F leads
  int32_t* cl_head;          // [N] head seed of merged cluster c
  int32_t *w0, *w1, *w2, *w3, *w4, *w5, *w6; // per-cluster scratch over the L index space
  // F: leads after merge_inner (fused), slot space = L index space; FI: final per-refined-cluster order -> F slot
  int32_t *F_orig, *F_svlen, *F_seq_len; int64_t* F_seq_off; uint8_t* F_sel; int32_t* FI;
  int32_t* F_lpos;           // L position (Lrec index) of the fused lead's head
  int32_t *rc_n_s, *rc_cl_s; uint8_t* rc_keeplong_s;   // slot space (sparse): slot = F position of the rc's first lead
  int32_t *rc_lo, *rc_n, *rc_cluster; uint8_t* rc_keeplong;              // dense
  snf_call_t* cand; CallX* candx;                                         // per rc
  snf_call_t* calls; CallX* callx;                                        // compacted
  uint32_t* rnames;          // [2N]

  // ---- stage E: finalize
  const GtEntry* gt_lut;     // [251*251]
  int32_t* cons_call;        // [n_cons] call index
  int64_t *cons_tab_off, *cons_aln_off, *cons_read_off, *cons_tab_sz;  // [n_cons] offsets (+ table slots) per ALT call
  int64_t *sz_tab, *sz_aln, *sz_rd;   // [n_calls+1] per-call sizes written by e2_best
  int64_t *sc_tab, *sc_aln, *sc_rd;   // [n_calls+1] their exclusive scans
  uint64_t* tab_key; int32_t* tab_pos; uint8_t* tab_state; int64_t tab_cap;   // anchor hash tables
  uint8_t* aln; int64_t aln_cap;         // aligned reads, n_others x L per consensus call
  uint8_t* aln_kept;         // [n_cons_reads]
  int32_t *cr_call, *cr_read;  // [n_cons_reads] (consensus id, other index)
  uint8_t* alt_pool; int64_t alt_cap;    // ALT bytes of all candidates, candidate order (HBM)
  uint8_t* alt_pin; int64_t alt_pin_cap; // the same section in pinned host memory (0: none); Counts::alt_in_pinned says which one a pass uses
  // staged result (another pass in flight: the kernels store into HBM): where z2_stage_copy takes the block / the ALT section (0: the fetch copies)
  uint8_t* stage_out_pin; int64_t stage_out_cap; uint8_t* stage_alt_pin; int64_t stage_alt_cap;
  unsigned long long* stripes;  // [4 classes][64 stripes][16] striped byte counters (one 128-B line each): cons_bytes
  unsigned long long* tile_super; int64_t super_stride;  // sums per 64 tiles (8 slots), zeroed at the start of a pass
  // single-launch "flags -> exclusive scan -> emit" chains (snf_fused.h chain_scan): per slot and 256-element tile two words
  // (tile aggregate, inclusive prefix), each tagged with the launch it belongs to; one ticket counter per slot
  unsigned long long* chain; int64_t chain_stride; uint32_t* chain_ticket; uint32_t* chain_epoch; int32_t chain_on, _pad_chain;
                             // chain_epoch: one word in HBM, bumped by z0_init at the start of every pass (a pass replayed from a HIP graph
                             // has no host-side counter to take its tags from)
  unsigned long long* tile_sums; int64_t tile_stride;  // per-256-element-tile sums of the fused size->scan->emit chains
  ConsDesc* cdesc;           // [n_cons] by cons id
#ifdef SNF_ITRACE
  uint32_t* itrace;   // measurement build only (-DSNF_ITRACE, tools/itrace.sh): per kernel slot and workgroup {start, duration} in 100 MHz ticks
#endif
#ifdef SNF_WG_TRACE
  unsigned long long* wgtrace;   // measurement build only (-DSNF_WG_TRACE): per consensus call {start, duration | shape} in 100 MHz ticks
#endif
  // refined clusters handed on by the grouped call kernels (snf_wave_call_g.h): list 0 more than 8 leads (d2g_call<8> -> <32>),
  // list 1 more than 32 (-> d2w_call).  64 stripes per list (stripe = workgroup & 63) with a counter each, d2cnt[(list * 64 +
  // stripe) * 16]: ten thousand returning atomics on ONE counter took 0.1 ms by themselves.  Stripe s owns d2_list[k][s * d2cap ...)
  // list 2: merged clusters of more than 8 leads, handed by d1g_refine<8> to d1w_refine (snf_wave_refine_g.h)
  int32_t* d2_list[3]; uint32_t* d2cnt; int64_t d2cap;
  int32_t d2_from_list, d1_from_list;   // != 0: this launch of d2w_call takes its refined clusters from d2_list[d2_from_list - 1] / d1w_refine its clusters from list 2
  int32_t* cls_list[8];      // cons ids per work list (see Counts::n_cls; 6 unused), appended with wave-aggregated atomics
  // clusters / refined clusters / calls with more than 64 leads, collected by the wave kernels (kind 0 d1w_refine, 1 d2w_call,
  // 2 e1w_finalize) in 64 stripes (item & 63) and served one wave each by x_big<kind> (snf_wave_call.h)
  uint32_t* big_cnt;         // [3][64][16]: one counter per stripe on its own 64-B line
  int32_t* big_list;         // [3][64][big_cap]
  int64_t big_cap;
  int32_t big_wave;          // 1: the thread kernels leave the items above to x_big
  int32_t e1_batch;          // calls per wave of e1w_finalize (SNF_E1_BATCH env: 2, 4, 8, 16, 32, 64; default 64)
  // w4s_segment in two launches: the 64-lead instance over every block (mode 1: a block whose last window does not fit appends itself to
  // w4_list and leaves), the large instance over that list (mode 2); mode 0: one launch of one instance
  int32_t* w4_list; int32_t w4_mode, _pad_w4;
  int32_t heavy_n;           // hand-over lists: an item with more leads than this goes into the first 16 stripes, i.e. to the FRONT of the index
                             // space the next kernel walks (its workgroups start in index order: the long items first, not last); 0: off
  int32_t wave_uniform;      // 1 (only inside x_big): the 64 lanes of the wave run the serial body in lock step; sorts are cooperative
  int32_t* w7;               // [N+1] scratch of the cooperative sorts (same slot space as w0..w6)
  // x_big<0> keeps a cluster in LDS: its packed lead records and the eight scratch rows (stage_cap entries each); null otherwise
  const LeadRec* stage_R; int32_t* stage_w; int32_t stage_cap; int32_t _pad_stage;
  uint8_t* aln_kept_w;       // [N+1] kept flag per (call, other read) of the workgroup kernels, indexed like crl_*
  int64_t* crl_off; int32_t* crl_len;  // [<= N] pool offset / length of every 'other' read, in cluster order per call
};

// measurement build (-DSNF_ITRACE): every workgroup of the kernels that carry IT_SCOPE(slot) leaves when it started and how long it
// ran - what the device's occupancy over a kernel's span looks like (ramp, plateau, tail) and which workgroups are the last
#define SNF_IT_SLOTS 16
#define SNF_IT_CAP (1 << 17)
#ifdef SNF_ITRACE
struct ItScope {
  uint32_t* o; unsigned long long t0;
  __device__ ItScope(const View& v, int slot) : o(nullptr), t0(wall_clock64()) {
    if (v.itrace && threadIdx.x == 0 && blockIdx.x < (unsigned)SNF_IT_CAP) o = v.itrace + 2 * ((int64_t)slot * SNF_IT_CAP + blockIdx.x);
  }
  __device__ ~ItScope() { if (o) { o[0] = (uint32_t)t0; o[1] = (uint32_t)(wall_clock64() - t0) | 0x80000000u; } }
};
#define IT_SCOPE(slot) ItScope it_scope_(v, slot);
#else
#define IT_SCOPE(slot)
#endif

}  // namespace snf
