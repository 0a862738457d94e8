/*
 * snf_oracle.c - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A plain, serial C restatement of the Sniffles2 clustering + consensus hot path
 * (fritzsedlazeck/Sniffles, pure Python).  It follows the reference control flow
 * statement by statement (lists, dict insertion order, sequential scans) and
 * shares NO code with the HIP implementation under sniffles_amd/csrc/.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinned against the reference itself: tests/golden/ fixtures are produced by
 * running the unmodified reference (oracle/ref_harness.py, oracle/make_golden.py)
 * and tests/test_oracle_golden.py requires this file to reproduce them exactly.
 *
 * Reference files followed (relative to the reference root):
 *   src/sniffles/leadprov.py:358-418,445-472   record_lead / record_hap_ref / coverage
 *   src/sniffles/cluster.py:27-353             Cluster, merge_inner, resplit, resplit_bnd, resolve
 *   src/sniffles/sv.py:484-639                 calculate_bounds, call_from, resolve_bnd
 *   src/sniffles/util.py:25-103,167            stdev, median_modes(center), trim, most_common
 *   src/sniffles/postprocessing.py:25-654      annotate_sv, coverage, qc_*, genotype_sv, phase_sv
 *   src/sniffles/genotyping.py:28-241          Genotyper + subclasses
 *   src/sniffles/consensus.py:142-144,280-394  iter_kmers, novel_from_reads
 *   src/sniffles/parallel.py:104-249           call_candidates, finalize_candidates, rescue_phasing
 */
#include "../include/sniffles_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ memory */
typedef struct Chunk { struct Chunk* next; size_t used, cap; char data[]; } Chunk;
typedef struct { Chunk* head; } Arena;
static void* arena_alloc(Arena* a, size_t n) {
  n = (n + 15) & ~(size_t)15;
  if (!a->head || a->head->used + n > a->head->cap) {
    size_t cap = n > (1u << 22) ? n : (1u << 22);
    Chunk* c = (Chunk*)malloc(sizeof(Chunk) + cap);
    if (!c) { fprintf(stderr, "snf_oracle: out of memory\n"); abort(); }
    c->next = a->head; c->used = 0; c->cap = cap; a->head = c;
  }
  void* p = a->head->data + a->head->used;
  a->head->used += n;
  return p;
}
static void arena_free(Arena* a) {
  Chunk* c = a->head;
  while (c) { Chunk* n = c->next; free(c); c = n; }
  a->head = NULL;
}

/* ------------------------------------------------------------------ lead */
typedef struct OLead {
  int32_t ref_start, ref_end, qry_start, qry_end;
  int64_t svlen; int has_svlen;
  int32_t read_len;
  uint32_t qname, read_id;
  int32_t ps;
  int32_t mate_contig, mate_ref_start;
  uint8_t svtype, strand, mapq, source, hap, is_sa, is_first, is_reverse;
  double nm;
  const uint8_t* seq; int64_t seq_len; /* seq == NULL <=> None */
  int64_t orig;
} OLead;

typedef struct { OLead** a; int64_t n, cap; } LVec;
static void lv_push(Arena* ar, LVec* v, OLead* l) {
  if (v->n == v->cap) {
    int64_t nc = v->cap ? v->cap * 2 : 8;
    OLead** na = (OLead**)arena_alloc(ar, (size_t)nc * sizeof(OLead*));
    if (v->n) memcpy(na, v->a, (size_t)v->n * sizeof(OLead*));
    v->a = na; v->cap = nc;
  }
  v->a[v->n++] = l;
}
static void lv_extend(Arena* ar, LVec* v, const LVec* o) {
  for (int64_t i = 0; i < o->n; i++) lv_push(ar, v, o->a[i]);
}

typedef struct OCluster {
  int32_t start, end, seed, seed_index;
  LVec leads;
  LVec leads_long; int has_long;
  int repeat;
  int32_t hap[6];
  double mean_svlen, stdev_start;
  int32_t sa_count; double sa_frac;
} OCluster;

/* ------------------------------------------------------------------ exact statistics */
/* correctly rounded double of num/den (exact rational), as float(Fraction) / int true division */
static double ratio_to_double(u128 num, u128 den) {
  if (num == 0) return 0.0;
  int bn = 0, bd = 0;
  for (u128 t = num; t; t >>= 1) bn++;
  for (u128 t = den; t; t >>= 1) bd++;
  int s = 55 - (bn - bd); /* quotient of (num<<s)/den has 55 or 56 bits */
  u128 N = num, D = den;
  if (s >= 0) N <<= s; else D <<= (-s);
  u128 q = N / D, r = N % D;
  if (r) q |= 1; /* sticky */
  double v = (double)(uint64_t)q; /* q < 2^57, conversion rounds to nearest even */
  return ldexp(v, -s);
}

/* statistics.stdev over integers (CPython 3.10): exact variance -> one rounding -> sqrt */
static double stdev_ints(const int64_t* x, int64_t n) {
  if (n < 2) return 0.0;
  int64_t x0 = x[0];
  u128 s2 = 0; __int128 s1 = 0;
  for (int64_t i = 0; i < n; i++) {
    __int128 d = (__int128)x[i] - x0;
    s1 += d; s2 += (u128)(d * d);
  }
  u128 num = (u128)n * s2 - (u128)(s1 * s1);
  u128 den = (u128)n * (u128)(n - 1);
  return sqrt(ratio_to_double(num, den));
}

static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* util.median_modes == center (util.py:49-58,167) */
static int64_t center_ints(Arena* ar, const int64_t* x, int64_t n) {
  int64_t* s = (int64_t*)arena_alloc(ar, (size_t)n * sizeof(int64_t));
  memcpy(s, x, (size_t)n * sizeof(int64_t));
  qsort(s, (size_t)n, sizeof(int64_t), cmp_i64);
  int64_t max_count = 0;
  for (int64_t i = 0; i < n;) { int64_t j = i; while (j < n && s[j] == s[i]) j++; if (j - i > max_count) max_count = j - i; i = j; }
  int64_t* keys = (int64_t*)arena_alloc(ar, (size_t)n * sizeof(int64_t));
  int64_t k = 0;
  for (int64_t i = 0; i < n;) { int64_t j = i; while (j < n && s[j] == s[i]) j++; if (max_count - (j - i) < 3) keys[k++] = s[i]; i = j; }
  return keys[k / 2]; /* median_noavg: sorted, index int(len/2) */
}

/* util.stdev(util.trim(nums)) (util.py:25-27,82-88) */
static double stdev_trim(Arena* ar, const int64_t* x, int64_t n) {
  int64_t* s = (int64_t*)arena_alloc(ar, (size_t)n * sizeof(int64_t));
  memcpy(s, x, (size_t)n * sizeof(int64_t));
  qsort(s, (size_t)n, sizeof(int64_t), cmp_i64);
  int64_t trim_n = (int64_t)((double)n / 100.0 * 25.0);
  if (trim_n > 0) return stdev_ints(s + trim_n, n - 2 * trim_n);
  return stdev_ints(s, n);
}

static int cmp_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x > y) - (x < y);
}
/* sorted distinct values; returns count */
static int64_t distinct_u32(uint32_t* v, int64_t n) {
  if (n == 0) return 0;
  qsort(v, (size_t)n, sizeof(uint32_t), cmp_u32);
  int64_t k = 1;
  for (int64_t i = 1; i < n; i++) if (v[i] != v[k - 1]) v[k++] = v[i];
  return k;
}

/* numpy pairwise summation of a contiguous float64 vector (np.sum / np.nanmean) */
static double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}
double snf_oracle_np_sum(const double* a, int64_t n) { return np_pairwise_sum(a, n); }
double snf_oracle_stdev(const int64_t* x, int64_t n) { return stdev_ints(x, n); }

/* ------------------------------------------------------------------ task state */
typedef struct OCall {
  snf_call_t c;
  OCluster* cluster;
  const uint8_t* alt; int64_t alt_len; /* alt != NULL: sequence */
  uint32_t* rn; int64_t rn_n;
} OCall;

typedef struct { OCall* a; int64_t n, cap; } CallVec;

typedef struct Task {
  Arena* ar;
  const snf_config_t* cfg;
  const snf_task_input_t* in;
  int task_index;
  OLead* leads; int64_t n;
  uint16_t* coverage;     /* dense uint16[contig_len] (leadprov.py:451,510) */
  uint16_t* hapref[3];    /* leadhapcount["REF"] per 100-bp bin (leadprov.py:387-398) */
  int64_t nbins;
  CallVec calls;
  int32_t sv_id;
  int status;
  double coverage_average_total;
} Task;

static void cv_push(CallVec* v, const OCall* c) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->a = (OCall*)realloc(v->a, (size_t)v->cap * sizeof(OCall)); }
  v->a[v->n++] = *c;
}

/* ------------------------------------------------------------------ cluster.py */
/* Cluster.compute_metrics (cluster.py:48-61) */
static void compute_metrics(Arena* ar, OCluster* c) {
  int64_t len = c->leads.n;
  int64_t n = len < 100 ? len : 100;
  if (n == 0) { c->mean_svlen = 0; c->stdev_start = 0; return; }
  int64_t step = len / n;
  if (n > 1) {
    int64_t cnt = 0; int64_t sum = 0;
    int64_t* xs = (int64_t*)arena_alloc(ar, (size_t)(len / step + 2) * sizeof(int64_t));
    for (int64_t i = 0; i < len; i += step) { sum += c->leads.a[i]->svlen; xs[cnt++] = c->leads.a[i]->ref_start; }
    c->mean_svlen = (double)sum / (double)n; /* divides by n, not by the sample count */
    c->stdev_start = stdev_ints(xs, cnt);
  } else {
    c->mean_svlen = (double)c->leads.a[0]->svlen;
    c->stdev_start = 0;
  }
}

typedef struct { int64_t first; int32_t ref_start; int64_t idx; OLead* l; } MIKey;
static int cmp_mikey(const void* a, const void* b) {
  const MIKey* x = (const MIKey*)a; const MIKey* y = (const MIKey*)b;
  if (x->first != y->first) return x->first < y->first ? -1 : 1;
  if (x->ref_start != y->ref_start) return x->ref_start < y->ref_start ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* merge_inner (cluster.py:85-122): mutates leads, regroups by read in first-appearance order */
static void merge_inner(Arena* ar, OCluster* c, int threshold) {
  int64_t n = c->leads.n;
  if (n == 0) return;
  MIKey* k = (MIKey*)arena_alloc(ar, (size_t)n * sizeof(MIKey));
  /* first appearance of each qname: open addressing */
  int64_t hs = 16; while (hs < 2 * n) hs <<= 1;
  int64_t* slot = (int64_t*)arena_alloc(ar, (size_t)hs * sizeof(int64_t));
  for (int64_t i = 0; i < hs; i++) slot[i] = -1;
  for (int64_t i = 0; i < n; i++) {
    OLead* l = c->leads.a[i];
    uint64_t h = ((uint64_t)l->qname * 0x9E3779B97F4A7C15ull) >> 20;
    int64_t p = (int64_t)(h & (uint64_t)(hs - 1));
    while (slot[p] >= 0 && c->leads.a[slot[p]]->qname != l->qname) p = (p + 1) & (hs - 1);
    if (slot[p] < 0) slot[p] = i;
    k[i].first = slot[p]; k[i].ref_start = l->ref_start; k[i].idx = i; k[i].l = l;
  }
  qsort(k, (size_t)n, sizeof(MIKey), cmp_mikey);
  LVec out = {0};
  int64_t i = 0;
  while (i < n) {
    int64_t j = i; while (j < n && k[j].first == k[i].first) j++;
    OLead* to_merge = k[i].l;
    OLead* curr = to_merge;
    int32_t last_ref_end = to_merge->ref_end, last_qry_end = to_merge->qry_end;
    int32_t last_ref_start = to_merge->ref_start, last_qry_start = to_merge->qry_start;
    for (int64_t t = i + 1; t < j; t++) {
      to_merge = k[t].l;
      int merge = (threshold == -1) ||
                  (((abs(to_merge->ref_start - last_ref_end) < threshold || abs(to_merge->ref_start - last_ref_start) < threshold) &&
                    (abs(to_merge->qry_start - last_qry_end) < threshold || abs(to_merge->qry_start - last_qry_start) < threshold)) &&
                   (curr->strand == to_merge->strand));
      if (merge) {
        curr->svlen += to_merge->svlen;
        if (to_merge->seq == NULL || curr->seq == NULL) { curr->seq = NULL; curr->seq_len = 0; }
        else {
          uint8_t* ns = (uint8_t*)arena_alloc(ar, (size_t)(curr->seq_len + to_merge->seq_len + 1));
          memcpy(ns, curr->seq, (size_t)curr->seq_len);
          memcpy(ns + curr->seq_len, to_merge->seq, (size_t)to_merge->seq_len);
          curr->seq = ns; curr->seq_len += to_merge->seq_len;
        }
      } else {
        lv_push(ar, &out, curr);
        curr = to_merge;
      }
      last_ref_end = to_merge->ref_end; last_qry_end = to_merge->qry_end;
      last_ref_start = to_merge->ref_start; last_qry_start = to_merge->qry_start;
    }
    lv_push(ar, &out, curr);
    i = j;
  }
  c->leads = out;
}

typedef struct { OCluster** a; int64_t n, cap; } ClVec;
static void clv_push(Arena* ar, ClVec* v, OCluster* c) {
  if (v->n == v->cap) {
    int64_t nc = v->cap ? v->cap * 2 : 16;
    OCluster** na = (OCluster**)arena_alloc(ar, (size_t)nc * sizeof(OCluster*));
    if (v->n) memcpy(na, v->a, (size_t)v->n * sizeof(OCluster*));
    v->a = na; v->cap = nc;
  }
  v->a[v->n++] = c;
}

static OCluster* cluster_derive(Arena* ar, const OCluster* c, LVec leads, int keep_long) {
  OCluster* nc = (OCluster*)arena_alloc(ar, sizeof(OCluster));
  *nc = *c;
  nc->leads = leads;
  if (!keep_long) { nc->has_long = 0; nc->leads_long.n = 0; nc->leads_long.a = NULL; nc->leads_long.cap = 0; }
  return nc;
}

typedef struct { int64_t key; LVec list; } BinList;
static int cmp_binlist(const void* a, const void* b) {
  int64_t x = ((const BinList*)a)->key, y = ((const BinList*)b)->key;
  return (x > y) - (x < y);
}

/* resplit (cluster.py:125-161), prop = lead.svlen */
static void resplit(Arena* ar, OCluster* c, ClVec* out, int binsize, int thr_min, double thr_frac) {
  int64_t n = c->leads.n;
  BinList* bins = (BinList*)arena_alloc(ar, (size_t)(n + 1) * sizeof(BinList));
  int64_t nb = 0;
  for (int64_t i = 0; i < n; i++) {
    OLead* l = c->leads.a[i];
    int64_t a = l->svlen < 0 ? -l->svlen : l->svlen;
    int64_t bin = (a / binsize) * binsize;
    int64_t b = 0;
    while (b < nb && bins[b].key != bin) b++;
    if (b == nb) { bins[nb].key = bin; memset(&bins[nb].list, 0, sizeof(LVec)); nb++; }
    lv_push(ar, &bins[b].list, l);
  }
  qsort(bins, (size_t)nb, sizeof(BinList), cmp_binlist);
  /* new_clusters = indices into bins[] (sorted keys) */
  int64_t* nc = (int64_t*)arena_alloc(ar, (size_t)(nb + 1) * sizeof(int64_t));
  int64_t m = nb;
  for (int64_t i = 0; i < nb; i++) nc[i] = i;
  int64_t i = 1;
  while (m > 1 && i < m) {
    int64_t im1 = (i == 0) ? m - 1 : i - 1; /* Python negative index wrap-around */
    int64_t last = bins[nc[im1]].key, curr = bins[nc[i]].key;
    int64_t mn = curr < last ? curr : last;
    double t = (double)mn * thr_frac;
    double thr = ((double)thr_min >= t) ? (double)thr_min : t; /* max(int, float) */
    int64_t diff = curr > last ? curr - last : last - curr;
    if ((double)diff <= thr) {
      lv_extend(ar, &bins[nc[i]].list, &bins[nc[im1]].list);
      for (int64_t t2 = im1; t2 + 1 < m; t2++) nc[t2] = nc[t2 + 1]; /* pop(i-1) */
      m--;
      i = (i - 2 > 0) ? i - 2 : 0;
    } else {
      i++;
    }
  }
  for (int64_t t2 = 0; t2 < m; t2++) clv_push(ar, out, cluster_derive(ar, c, bins[nc[t2]].list, 1));
}

typedef struct { int32_t contig; int is_first; BinList* bins; int64_t nb; } BndIdent;

/* resplit_bnd (cluster.py:164-216) */
static void resplit_bnd(Arena* ar, OCluster* c, ClVec* out, int thr) {
  int64_t n = c->leads.n;
  if (n <= 1) { clv_push(ar, out, c); return; }
  BndIdent* ids = (BndIdent*)arena_alloc(ar, (size_t)n * sizeof(BndIdent));
  int64_t nid = 0;
  for (int64_t i = 0; i < n; i++) {
    OLead* l = c->leads.a[i];
    int64_t d = 0;
    while (d < nid && !(ids[d].contig == l->mate_contig && ids[d].is_first == l->is_first)) d++;
    if (d == nid) { ids[nid].contig = l->mate_contig; ids[nid].is_first = l->is_first; ids[nid].bins = (BinList*)arena_alloc(ar, (size_t)n * sizeof(BinList)); ids[nid].nb = 0; nid++; }
    int64_t pos_bin = thr > 0 ? ((int64_t)l->mate_ref_start / thr) * thr : 0;
    BndIdent* I = &ids[d];
    int64_t b = 0;
    while (b < I->nb && I->bins[b].key != pos_bin) b++;
    if (b == I->nb) { I->bins[b].key = pos_bin; memset(&I->bins[b].list, 0, sizeof(LVec)); I->nb++; }
    lv_push(ar, &I->bins[b].list, l);
  }
  for (int64_t d = 0; d < nid; d++) {
    BndIdent* I = &ids[d];
    qsort(I->bins, (size_t)I->nb, sizeof(BinList), cmp_binlist);
    LVec curr = {0};
    lv_extend(ar, &curr, &I->bins[0].list);
    int64_t last_bin = I->bins[0].key;
    for (int64_t b = 1; b < I->nb; b++) {
      int64_t pb = I->bins[b].key;
      if (pb - last_bin <= thr) {
        lv_extend(ar, &curr, &I->bins[b].list);
      } else {
        if (curr.n) clv_push(ar, out, cluster_derive(ar, c, curr, 0));
        memset(&curr, 0, sizeof(curr));
        lv_extend(ar, &curr, &I->bins[b].list);
      }
      last_bin = pb;
    }
    if (curr.n) clv_push(ar, out, cluster_derive(ar, c, curr, 0));
  }
}

typedef struct { int64_t key; int64_t idx; } SortKey;
static int cmp_sortkey(const void* a, const void* b) {
  const SortKey* x = (const SortKey*)a; const SortKey* y = (const SortKey*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* cluster.resolve (cluster.py:219-353) for one svtype; appends refined clusters to out */
static void resolve(Task* T, int svtype, ClVec* out) {
  Arena* ar = T->ar; const snf_config_t* cfg = T->cfg;
  int binsize = cfg->cluster_binsize;
  /* leadtab[svtype]: bins in sorted order, each list in arrival order (leadprov.py:400-418) */
  int64_t cnt = 0;
  for (int64_t i = 0; i < T->n; i++) if (T->leads[i].svtype == svtype && T->leads[i].orig >= 0) cnt++;
  if (cnt == 0) return;
  SortKey* sk = (SortKey*)arena_alloc(ar, (size_t)cnt * sizeof(SortKey));
  int64_t m = 0;
  for (int64_t i = 0; i < T->n; i++) {
    OLead* l = &T->leads[i];
    if (l->svtype != svtype || l->orig < 0) continue;
    sk[m].key = ((int64_t)l->ref_start / binsize) * binsize; sk[m].idx = i; m++;
  }
  qsort(sk, (size_t)cnt, sizeof(SortKey), cmp_sortkey);

  int64_t n_tr = T->in->n_tr; /* <0: None; 0: empty list -> None (cluster.py:232-233) */
  int have_tr = n_tr > 0;
  int64_t tr_index = 0; int32_t tr_start = 0, tr_end = 0;
  if (have_tr) { tr_start = T->in->tr_start[0]; tr_end = T->in->tr_end[0]; }

  ClVec clusters = {0};
  int32_t seed_index = 0;
  for (int64_t i = 0; i < cnt; seed_index++) {
    int64_t j = i; while (j < cnt && sk[j].key == sk[i].key) j++;
    int32_t seed = (int32_t)sk[i].key;
    int within_tr = 0;
    if (have_tr && tr_index < n_tr) {
      while (tr_end < seed && tr_index + 1 < n_tr) { tr_index++; tr_start = T->in->tr_start[tr_index]; tr_end = T->in->tr_end[tr_index]; }
      if (tr_start < seed && seed < tr_end) within_tr = 1;
    }
    /* record_lead side effects for this bin: seq cap + hap counters */
    int32_t hapc[3] = {0, 0, 0};
    for (int64_t t = i; t < j; t++) {
      OLead* l = &T->leads[sk[t].idx];
      if (t - i + 1 > cfg->consensus_max_reads_bin) { l->seq = NULL; l->seq_len = 0; }
      if (hapc[l->hap] < 65535) hapc[l->hap]++; /* array('H') OverflowError -> not incremented */
    }
    LVec leads = {0}, leads_long = {0};
    for (int64_t t = i; t < j; t++) {
      OLead* l = &T->leads[sk[t].idx];
      if (svtype == SNF_INS && !l->has_svlen) lv_push(ar, &leads_long, l); else lv_push(ar, &leads, l);
    }
    if (leads.n >= cfg->dev_min_leads_cluster) {
      OCluster* c = (OCluster*)arena_alloc(ar, sizeof(OCluster));
      memset(c, 0, sizeof(*c));
      c->start = seed; c->end = seed + binsize; c->seed = seed; c->seed_index = seed_index;
      c->leads = leads; c->leads_long = leads_long; c->has_long = (svtype == SNF_INS);
      c->repeat = within_tr || cfg->repeat;
      c->hap[0] = hapc[0]; c->hap[1] = hapc[1]; c->hap[2] = hapc[2];
      int64_t b = seed / binsize;
      if (b >= 0 && b < T->nbins) { c->hap[3] = T->hapref[0][b]; c->hap[4] = T->hapref[1][b]; c->hap[5] = T->hapref[2][b]; }
      compute_metrics(ar, c);
      clv_push(ar, &clusters, c);
    }
    i = j;
  }

  /* adaptive merge scan (cluster.py:278-308) */
  int64_t i = 0;
  while (i < clusters.n - 1) {
    OCluster* curr = clusters.a[i]; OCluster* next = clusters.a[i + 1];
    int64_t inner = (int64_t)next->start - curr->end;
    int64_t outer = (int64_t)next->end - curr->start;
    double ms = curr->stdev_start < next->stdev_start ? curr->stdev_start : next->stdev_start; /* min(a,b) */
    int merge = (double)inner <= ms * cfg->cluster_r;
    if (!merge && (cfg->repeat || curr->repeat || next->repeat)) {
      double h = (fabs(curr->mean_svlen) + fabs(next->mean_svlen)) * cfg->cluster_repeat_h;
      double lim = h < cfg->cluster_repeat_h_max ? h : cfg->cluster_repeat_h_max; /* min(h_max, h) */
      merge = (double)outer <= lim;
    }
    if (!merge && svtype == SNF_BND) merge = inner <= cfg->cluster_merge_bnd;
    if (merge) {
      for (int64_t t = i + 1; t + 1 < clusters.n; t++) clusters.a[t] = clusters.a[t + 1];
      clusters.n--;
      lv_extend(ar, &curr->leads, &next->leads);
      if (svtype == SNF_INS) lv_extend(ar, &curr->leads_long, &next->leads_long);
      curr->end = next->end;
      curr->repeat = curr->repeat || next->repeat;
      compute_metrics(ar, curr);
      i = (i - 2 > 0) ? i - 2 : 0;
    }
    i++;
  }

  for (int64_t ci = 0; ci < clusters.n; ci++) {
    OCluster* c = clusters.a[ci];
    if (c->leads.n == 0) continue;
    if (svtype == SNF_BND) {
      if (cfg->dev_no_resplit) clv_push(ar, out, c);
      else resplit_bnd(ar, c, out, cfg->cluster_merge_bnd);
    } else {
      if (svtype == SNF_INS || svtype == SNF_DEL) merge_inner(ar, c, c->repeat ? -1 : cfg->cluster_merge_pos);
      if (!cfg->dev_no_resplit_repeat && !cfg->dev_no_resplit)
        resplit(ar, c, out, cfg->cluster_resplit_binsize, cfg->minsvlen, cfg->cluster_merge_len);
      else clv_push(ar, out, c);
    }
  }
}

/* ------------------------------------------------------------------ sv.py call_from */
static int64_t collect_qnames(Arena* ar, const LVec* v, uint32_t** outp) {
  uint32_t* q = (uint32_t*)arena_alloc(ar, (size_t)(v->n + 1) * sizeof(uint32_t));
  for (int64_t i = 0; i < v->n; i++) q[i] = v->a[i]->qname;
  *outp = q;
  return distinct_u32(q, v->n);
}

/* util.most_common_top over small integer values: max count, ties -> smallest value */
static int32_t most_common_top_i32(Arena* ar, const int32_t* v, int64_t n) {
  int64_t* s = (int64_t*)arena_alloc(ar, (size_t)n * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) s[i] = v[i];
  qsort(s, (size_t)n, sizeof(int64_t), cmp_i64);
  int64_t best = s[0], bc = 0;
  for (int64_t i = 0; i < n;) { int64_t j = i; while (j < n && s[j] == s[i]) j++; if (j - i > bc) { bc = j - i; best = s[i]; } i = j; }
  return (int32_t)best;
}

static void call_from(Task* T, OCluster* cl, int svtype) {
  Arena* ar = T->ar; const snf_config_t* cfg = T->cfg;
  LVec* leads = &cl->leads;
  int64_t n = leads->n;
  int64_t* tmp = (int64_t*)arena_alloc(ar, (size_t)n * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) tmp[i] = leads->a[i]->svlen;
  int64_t svlen = center_ints(ar, tmp, n);
  int is_single = (svtype == SNF_SINGLE_LEFT || svtype == SNF_SINGLE_RIGHT);
  if (!is_single && svtype != SNF_BND) {
    int64_t a = svlen < 0 ? -svlen : svlen;
    if (a < cfg->minsvlen_screen) return;
  }
  uint32_t* q; int64_t nq = collect_qnames(ar, leads, &q);
  int64_t support, support_long = 0;
  uint32_t* rn = q; int64_t rn_n = nq;
  if (svtype == SNF_INS && svlen >= cfg->long_ins_length) {
    uint32_t* ql; int64_t nql = collect_qnames(ar, &cl->leads_long, &ql);
    support_long = nql;
    uint32_t* u = (uint32_t*)arena_alloc(ar, (size_t)(nq + nql + 1) * sizeof(uint32_t));
    memcpy(u, q, (size_t)nq * sizeof(uint32_t)); memcpy(u + nq, ql, (size_t)nql * sizeof(uint32_t));
    rn_n = distinct_u32(u, nq + nql); rn = u;
    support = rn_n;
  } else support = nq;
  for (int64_t i = 0; i < n; i++) tmp[i] = leads->a[i]->ref_start;
  int64_t ref_start = center_ints(ar, tmp, n);
  double stdev_pos = stdev_trim(ar, tmp, n);
  double stdev_len = NAN; int precise;
  if (svtype != SNF_BND) {
    for (int64_t i = 0; i < n; i++) tmp[i] = leads->a[i]->svlen;
    stdev_len = stdev_trim(ar, tmp, n);
    precise = (stdev_pos + stdev_len < (double)cfg->precise);
  } else precise = (stdev_pos < (double)cfg->precise);
  int64_t svstart, svend;
  if (svtype == SNF_INS) { svstart = ref_start; svend = ref_start; }
  else if (svtype == SNF_DEL) { svstart = ref_start + svlen; svend = ref_start; }
  else { svstart = ref_start; svend = svstart + (svlen < 0 ? -svlen : svlen); }
  int64_t msum = 0; for (int64_t i = 0; i < n; i++) msum += leads->a[i]->mapq;
  int qual = (int)((double)msum / (double)n);
  int64_t fwd = 0; for (int64_t i = 0; i < n; i++) fwd += (leads->a[i]->strand == 0);
  double nm_mean = -1;
  if (cfg->qc_nm_measure) { double s = 0; for (int64_t i = 0; i < n; i++) s += leads->a[i]->nm; nm_mean = s / (double)n; }

  OCall oc; memset(&oc, 0, sizeof(oc));
  snf_call_t* c = &oc.c;
  c->task_index = T->task_index; c->sv_id = T->sv_id; c->svtype = svtype;
  c->pos = (int32_t)svstart; c->end = (int32_t)svend; c->svlen = (int32_t)svlen;
  c->support = (int32_t)support; c->support_long = -1; c->support_sa = -1;
  c->qual = qual; c->precise = precise; c->fwd = (int32_t)fwd; c->rev = (int32_t)(n - fwd);
  c->qc = 1; c->filter = SNF_F_PASS; c->nm = nm_mean;
  c->stdev_pos = stdev_pos; c->stdev_len = stdev_len;
  c->sa_count = cl->sa_count; c->sa_frac = cl->sa_frac;
  c->mate_contig = -1; c->gt_hp = -1; c->gt_ps = -1; c->vaf = NAN; c->alt_len = -1;
  c->cluster_start = cl->start; c->cluster_end = cl->end; c->cluster_seed_index = cl->seed_index;
  oc.cluster = cl; oc.rn = rn; oc.rn_n = rn_n;

  if (svtype == SNF_BND) { /* resolve_bnd (sv.py:625-639) */
    int32_t* v = (int32_t*)arena_alloc(ar, (size_t)n * sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) v[i] = leads->a[i]->mate_contig;
    int32_t mc = most_common_top_i32(ar, v, n);
    LVec sel = {0};
    for (int64_t i = 0; i < n; i++) if (leads->a[i]->mate_contig == mc) lv_push(ar, &sel, leads->a[i]);
    for (int64_t i = 0; i < sel.n; i++) tmp[i] = sel.a[i]->mate_ref_start;
    int64_t mrs = center_ints(ar, tmp, sel.n);
    for (int64_t i = 0; i < sel.n; i++) v[i] = sel.a[i]->is_first;
    int is_first = most_common_top_i32(ar, v, sel.n);
    for (int64_t i = 0; i < sel.n; i++) v[i] = sel.a[i]->is_reverse;
    int is_reverse = most_common_top_i32(ar, v, sel.n);
    uint32_t* qs; int64_t nqs = collect_qnames(ar, &sel, &qs);
    c->support = (int32_t)nqs;
    cl->leads = sel;
    c->mate_contig = mc; c->mate_ref_start = (int32_t)mrs; c->bnd_is_first = is_first; c->bnd_is_reverse = is_reverse;
  } else if (svtype == SNF_INS) {
    c->support_long = (int32_t)support_long;
  } else if (svtype == SNF_DEL) {
    int32_t s = 0; for (int64_t i = 0; i < n; i++) s += (leads->a[i]->source != SNF_SRC_INLINE);
    c->support_sa = s;
  }
  c->n_leads = (int32_t)cl->leads.n;
  T->sv_id++;
  cv_push(&T->calls, &oc);
}

/* ------------------------------------------------------------------ postprocessing.coverage */
static int cv_get(const Task* T, int64_t idx, int32_t* out) {
  int64_t len = T->in->contig_len;
  if (idx < -len || idx >= len) return 0; /* IndexError: field keeps its value */
  if (idx < 0) idx += len;                /* numpy negative index */
  *out = T->coverage[idx];
  return 1;
}

static void annotate_coverage(Task* T) {
  const snf_config_t* cfg = T->cfg;
  int bs = cfg->coverage_binsize, ud = cfg->coverage_updown_bins;
  int have_end = 0; int64_t end = 0;
  for (int64_t i = 0; i < T->calls.n; i++) {
    snf_call_t* c = &T->calls.a[i].c;
    int64_t start = c->pos;
    if (c->svtype == SNF_INS) { end = start + 1; have_end = 1; }
    else if (c->svtype == SNF_BND) { if (c->bnd_is_first) start -= 1; }
    else { end = (int64_t)c->pos + (c->svlen < 0 ? -(int64_t)c->svlen : c->svlen); have_end = 1; }
    if (!have_end) { T->status = SNF_TASK_ERR_UNBOUND_END; return; } /* UnboundLocalError */
    if (c->svtype == SNF_INS || c->svtype == SNF_BND) {
      cv_get(T, start - bs, &c->cov[1]);
      cv_get(T, start, &c->cov[2]);
      cv_get(T, end + bs, &c->cov[3]);
    } else {
      cv_get(T, start, &c->cov[1]);
      cv_get(T, (start + end) / 2, &c->cov[2]);
      cv_get(T, end - bs, &c->cov[3]);
    }
    cv_get(T, start - (int64_t)bs * ud, &c->cov[0]);
    cv_get(T, end + (int64_t)bs * ud, &c->cov[4]);
  }
}

/* ------------------------------------------------------------------ QC (postprocessing.py) */
static int64_t iabs64(int64_t x) { return x < 0 ? -x : x; }
static double py_round(double x) { return nearbyint(x); } /* round-half-even, default FP mode */

static int64_t rescale_support(const snf_call_t* c, const snf_config_t* cfg) {
  if (c->svtype != SNF_INS || c->svlen < cfg->long_ins_length) return c->support;
  double scale = cfg->long_ins_rescale_mult * ((double)c->svlen / (double)cfg->long_ins_length);
  return (int64_t)py_round((double)c->support * (cfg->long_ins_rescale_base + scale));
}

static int qc_support_auto(const snf_call_t* c, double cov_global, const snf_config_t* cfg) {
  int64_t support = rescale_support(c, cfg);
  int64_t lst[3]; int k = 0;
  if (c->cov[0] != 0) lst[k++] = c->cov[0];
  if (c->cov[4] != 0) lst[k++] = c->cov[4];
  if (k == 0) { for (int j = 1; j <= 3; j++) if (c->cov[j] != 0) lst[k++] = c->cov[j]; }
  double regional;
  if (k == 0) regional = cov_global;
  else {
    int64_t s = 0; for (int j = 0; j < k; j++) s += lst[j];
    regional = py_round((double)s / (double)k);
    if (regional == 0) regional = cov_global;
  }
  double gw = 1.0 - cfg->minsupport_auto_regional_coverage_weight;
  double cov = regional * cfg->minsupport_auto_regional_coverage_weight + cov_global * gw;
  double min_support = py_round(cfg->minsupport_auto_base + cfg->minsupport_auto_mult * cov);
  return (double)support >= min_support;
}

static int qc_sv_support(snf_call_t* c, double cov_global, const snf_config_t* cfg) {
  int ok = (cfg->minsupport < 0) ? qc_support_auto(c, cov_global, cfg) : (c->support >= cfg->minsupport);
  if (!ok) { c->filter = SNF_F_SUPPORT_MIN; return 0; }
  return 1;
}

static int distinct_strands(const OCluster* cl) {
  int f = 0, r = 0;
  for (int64_t i = 0; i < cl->leads.n; i++) { if (cl->leads.a[i]->strand == 0) f = 1; else r = 1; }
  return f + r;
}

static int qc_sv(OCall* oc, const snf_config_t* cfg) {
  snf_call_t* c = &oc->c;
  int t = c->svtype;
  int single = (t == SNF_SINGLE_LEFT || t == SNF_SINGLE_RIGHT);
  double alen = (double)iabs64(c->svlen);
  if (cfg->qc_stdev) {
    if (c->stdev_pos > (double)cfg->qc_stdev_abs_max) { c->filter = SNF_F_STDEV_POS; return 0; }
    if (t != SNF_BND && !single && c->stdev_pos / alen > 2.0) { c->filter = SNF_F_STDEV_POS; return 0; }
    if (!isnan(c->stdev_len) && c->stdev_len != 0) {
      if (t != SNF_BND && c->stdev_len / alen > 1.0) { c->filter = SNF_F_STDEV_LEN; return 0; }
      if (c->stdev_len > (double)cfg->qc_stdev_abs_max) { c->filter = SNF_F_STDEV_LEN; return 0; }
    }
  }
  if (single && !cfg->dev_output_candidates) { c->filter = SNF_F_SINGLE_BREAK; return 0; }
  if (iabs64(c->svlen) < cfg->minsvlen && t != SNF_BND) {
    if (c->support < 10 || cfg->minsvlen_hard_cap) { c->filter = SNF_F_SVLEN_MIN; return 0; }
  }
  if (t == SNF_BND) {
    if (cfg->qc_bnd_filter_strand && distinct_strands(oc->cluster) < 2) { c->filter = SNF_F_STRAND_BND; return 0; }
  }
  double up = c->cov[0], st = c->cov[1], ce = c->cov[2], en = c->cov[3], dn = c->cov[4];
  (void)st; (void)en;
  if (t == SNF_DEL && cfg->long_del_length != -1 && iabs64(c->svlen) >= cfg->long_del_length && !cfg->mosaic &&
      iabs64(c->svlen) <= cfg->dev_longer_del) {
    double scaled = cfg->long_del_coverage / 2.0;
    if (ce > (up + dn) * scaled) {
      if (up > ce && ce > dn) { if (dn / up < 0.7) { c->filter = SNF_F_COV_CHANGE_DEL; return 0; } }
      else if (up < ce && ce < dn) { if (up / dn < 0.7) { c->filter = SNF_F_COV_CHANGE_DEL; return 0; } }
    }
    if (up > dn) { if (0.5 > dn / up || ce > dn) { c->filter = SNF_F_COV_CHANGE_DEL; return 0; } }
    else if (up < dn) { if (0.5 > up / dn || up < ce) { c->filter = SNF_F_COV_CHANGE_DEL; return 0; } }
  } else if (t == SNF_DUP && cfg->long_dup_length != -1 && iabs64(c->svlen) >= cfg->long_dup_length && !cfg->mosaic &&
             iabs64(c->svlen) <= cfg->dev_longer_dup) {
    double scaled = cfg->long_dup_coverage / 2.0;
    if (ce < (up + dn) * scaled) {
      if (up > ce && ce > dn) { if (dn / up < 0.7) { c->filter = SNF_F_COV_CHANGE_DUP; return 0; } }
      else if (up < ce && ce < dn) { if (up / dn < 0.7) { c->filter = SNF_F_COV_CHANGE_DUP; return 0; } }
      if (up > dn) { if (0.5 > dn / up || ce < dn) { c->filter = SNF_F_COV_CHANGE_DUP; return 0; } }
      else if (up < dn) { if (0.5 > up / dn || up > ce) { c->filter = SNF_F_COV_CHANGE_DUP; return 0; } }
    }
  } else if (t == SNF_INS && (c->cov[0] < cfg->qc_coverage || c->cov[4] < cfg->qc_coverage)) {
    c->filter = SNF_F_COV_CHANGE_INS; return 0;
  }
  if (t == SNF_INS || t == SNF_DEL) {
    int no_split_sa = (c->support_sa <= 0); /* None or 0 */
    if (c->sa_frac > cfg->dev_inline_sa_support_max && c->sa_count > 5 && no_split_sa) { c->filter = SNF_F_INLINE_SA; return 0; }
  }
  /* qc_coverage_samples(): the sampler is never pushed to -> always (True, None) (sv.py:219-223) */
  double f = cfg->qc_coverage_max_change_frac;
  if (f != -1.0) {
    double u = c->cov[0] ? c->cov[0] : 1.0, s = c->cov[1] ? c->cov[1] : 1.0, m = c->cov[2] ? c->cov[2] : 1.0,
           e = c->cov[3] ? c->cov[3] : 1.0, d = c->cov[4] ? c->cov[4] : 1.0;
    if (fabs(u - s) / fmax(u, s) > f) { c->filter = SNF_F_COV_CHANGE_FRAC_US; return 0; }
    if (fabs(s - m) / fmax(s, m) > f) { c->filter = SNF_F_COV_CHANGE_FRAC_SC; return 0; }
    if (fabs(m - e) / fmax(m, e) > f) { c->filter = SNF_F_COV_CHANGE_FRAC_CE; return 0; }
    if (fabs(e - d) / fmax(e, d) > f) { c->filter = SNF_F_COV_CHANGE_FRAC_ED; return 0; }
  }
  return 1;
}

/* phase_sv (postprocessing.py:626-654) */
typedef struct { int32_t val; int64_t cnt; } ValCnt;
static int cmp_valcnt_desc(const void* a, const void* b) { /* sorted((count, value), reverse=True) */
  const ValCnt* x = (const ValCnt*)a; const ValCnt* y = (const ValCnt*)b;
  if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;
  return (x->val < y->val) - (x->val > y->val);
}
#define PS_NULL_CODE 0x7fffffff

static void phase_sv(Task* T, OCall* oc, int* hp_ret, int* ps_ret) {
  Arena* ar = T->ar; const snf_config_t* cfg = T->cfg;
  OCluster* cl = oc->cluster; snf_call_t* c = &oc->c;
  int64_t n = cl->leads.n;
  /* reads_phases = {read_id: (hap, ps)}: last lead of a read wins */
  typedef struct { uint32_t rid; int64_t idx; } RK;
  RK* rk = (RK*)arena_alloc(ar, (size_t)n * sizeof(RK));
  int64_t nr = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t d = 0; while (d < nr && rk[d].rid != cl->leads.a[i]->read_id) d++;
    if (d == nr) { rk[nr].rid = cl->leads.a[i]->read_id; nr++; }
    rk[d].idx = i;
  }
  ValCnt* hp = (ValCnt*)arena_alloc(ar, (size_t)(nr + 1) * sizeof(ValCnt));
  ValCnt* ps = (ValCnt*)arena_alloc(ar, (size_t)(nr + 1) * sizeof(ValCnt));
  int64_t nh = 0, np_ = 0;
  for (int64_t r = 0; r < nr; r++) {
    OLead* l = cl->leads.a[rk[r].idx];
    int32_t h = l->hap;
    int32_t p = (l->ps == SNF_PS_NONE || l->ps == T->in->ps_null_rank) ? PS_NULL_CODE : l->ps;
    int64_t d = 0; while (d < nh && hp[d].val != h) d++;
    if (d == nh) { hp[nh].val = h; hp[nh].cnt = 0; nh++; }
    hp[d].cnt++;
    d = 0; while (d < np_ && ps[d].val != p) d++;
    if (d == np_) { ps[np_].val = p; ps[np_].cnt = 0; np_++; }
    ps[d].cnt++;
  }
  qsort(hp, (size_t)nh, sizeof(ValCnt), cmp_valcnt_desc);
  qsort(ps, (size_t)np_, sizeof(ValCnt), cmp_valcnt_desc);
  int64_t hp_support = hp[0].cnt, ps_support = ps[0].cnt;
  int32_t hpv = hp[0].val, psv = ps[0].val;
  int64_t other_hp = 0, other_ps = 0;
  for (int64_t i = 0; i < nh; i++) if (hp[i].val != hpv) other_hp += hp[i].cnt; /* hap strings are never "NULL" */
  for (int64_t i = 0; i < np_; i++) if (ps[i].val != psv && ps[i].val != PS_NULL_CODE) other_ps += ps[i].cnt;
  int hp_pass = ((double)other_hp / (double)(hp_support + other_hp) < cfg->phase_conflict_threshold) && hp_support > 0;
  int ps_pass = ((double)other_ps / (double)(ps_support + other_ps) < cfg->phase_conflict_threshold) && psv != PS_NULL_CODE && ps_support > 0;
  c->ph_set = 1; c->ph_hp = hpv; c->ph_ps = (psv == PS_NULL_CODE) ? -2 : psv;
  c->ph_hp_support = (int32_t)hp_support; c->ph_ps_support = (int32_t)ps_support;
  c->ph_hp_pass = hp_pass; c->ph_ps_pass = ps_pass;
  *hp_ret = ((hpv == 1 || hpv == 2) && hp_pass) ? hpv : -1;
  *ps_ret = ps_pass ? psv : -1;
}

/* genotyping.Genotyper.calculate (+ subclasses) and postprocessing.genotype_sv */
static int coverage_from_list(const int64_t* lst, int k, int64_t* out) {
  int64_t s = 0; int m = 0;
  for (int j = 0; j < k; j++) if (lst[j] != 0) { s += lst[j]; m++; }
  if (m == 0) return 0; /* UnknownGenotypeError */
  *out = (int64_t)py_round((double)s / (double)m);
  return 1;
}

static double likelihood_ratio(double q1, double q2) {
  if (q1 / q2 > 0) return log(q1 / q2) / log(10.0); /* math.log(x, 10) */
  return 0;
}

static void genotype_sv(OCall* oc, const snf_config_t* cfg, int hp_ret, int ps_ret) {
  snf_call_t* c = &oc->c;
  int t = c->svtype;
  int64_t support = (t == SNF_INS) ? rescale_support(c, cfg) : c->support;
  int64_t coverage = 0; int ok;
  int64_t l3[3];
  if (t == SNF_INS) { l3[0] = c->cov[2]; ok = coverage_from_list(l3, 1, &coverage); }
  else if (t == SNF_DEL) {
    int64_t sa = c->support_sa > 0 ? c->support_sa : 0;
    l3[0] = c->cov[1] + sa; l3[1] = c->cov[2] + sa; l3[2] = c->cov[3] + sa; ok = coverage_from_list(l3, 3, &coverage);
  } else if (t == SNF_DUP) {
    l3[0] = c->cov[1]; l3[1] = c->cov[3]; ok = coverage_from_list(l3, 2, &coverage);
    if (ok) coverage += (int64_t)py_round((double)support * 0.75);
  } else if (t == SNF_INV) {
    l3[0] = c->cov[0]; l3[1] = c->cov[4]; ok = coverage_from_list(l3, 2, &coverage);
    if (ok) coverage += (int64_t)py_round((double)support * 0.5);
  } else { l3[0] = c->cov[1]; l3[1] = c->cov[2]; l3[2] = c->cov[3]; ok = coverage_from_list(l3, 3, &coverage); }
  if (!ok) { c->filter = SNF_F_GT_FAILED; c->qc = 0; return; }
  if (support > coverage) coverage = support;
  double af = (double)support / (double)coverage;
  double p[3] = {cfg->genotype_error, 1.0 / (double)cfg->genotype_ploidy, 1.0 - cfg->genotype_error};
  int64_t max_lead = support > coverage ? support : coverage;
  int64_t ns = support, ncv = coverage;
  if (max_lead > 250) {
    double norm = 250.0 / (double)max_lead;
    ns = (int64_t)py_round((double)support * norm);
    ncv = (int64_t)py_round((double)coverage * norm);
  }
  double q[3]; int order[3] = {0, 1, 2};
  for (int g = 0; g < 3; g++) q[g] = pow(p[g], (double)ns) * pow(1.0 - p[g], (double)(ncv - ns));
  /* stable sort by q descending */
  for (int a = 1; a < 3; a++) { int o = order[a]; int b = a - 1; while (b >= 0 && q[order[b]] < q[o]) { order[b + 1] = order[b]; b--; } order[b + 1] = o; }
  double sum = 0; for (int g = 0; g < 3; g++) sum += q[order[g]];
  double nq[3]; for (int g = 0; g < 3; g++) nq[g] = q[order[g]] / sum;
  double q1 = nq[0], q2 = nq[1], qz = 0;
  for (int g = 0; g < 3; g++) if (order[g] == 0) { qz = nq[g]; break; }
  int64_t z = (int64_t)((-10.0) * likelihood_ratio(qz, q1)); if (z > 60) z = 60;
  int64_t gq = (int64_t)((-10.0) * likelihood_ratio(q2, q1)); if (gq > 60) gq = 60;
  int update_this_dup = (t == SNF_DUP) && af >= cfg->dev_min_dup_vaf;
  int flt = (z < cfg->genotype_min_z_score) && !cfg->mosaic;
  if (t == SNF_INS && flt && c->svlen >= cfg->long_ins_length && cfg->detect_large_ins) flt = 0;
  if (c->filter == SNF_F_PASS && flt) {
    c->filter = update_this_dup ? SNF_F_PASS : SNF_F_GT;
    c->qc = !cfg->pass_only;
  }
  static const int GA[3] = {0, 0, 1}, GB[3] = {0, 1, 1};
  int a = GA[order[0]], b = GB[order[0]];
  if (update_this_dup && order[0] == 0) { a = 0; b = 1; }
  c->gt_set = 1; c->gt_a = a; c->gt_b = b; c->gt_gq = (int32_t)gq;
  c->gt_dr = (int32_t)(coverage - support); c->gt_dv = (int32_t)support;
  c->gt_hp = hp_ret; c->gt_ps = ps_ret;
  c->vaf = af;
  /* post haplotype assessment (postprocessing.py:612-623) */
  if (a == 1 && b == 1 && c->ph_set) {
    if (c->ph_hp != 0) { c->ph_hp_pass = 1; c->gt_hp = c->ph_hp; c->gt_ps = c->ph_ps; }
  }
}

/* ------------------------------------------------------------------ consensus.py */
typedef struct { uint64_t key; int32_t pos; int32_t state; } KSlot; /* state: 0 empty, 1 anchor, 2 taboo */

static uint64_t kmer_key(const uint8_t* s, int klen) {
  uint64_t k = 0;
  for (int i = 0; i < klen; i++) k = (k << 8) | s[i];
  return k;
}

static void most_common2(const uint8_t* v, int64_t n, int64_t* c0, uint8_t* ch0, int64_t* c1, int* ndist) {
  /* util.most_common: sorted((count, char), reverse=True); returns top two counts and top char */
  int64_t cnt[256]; memset(cnt, 0, sizeof(cnt));
  for (int64_t i = 0; i < n; i++) cnt[v[i]]++;
  int64_t b0 = -1, b1 = -1; int k0 = -1, k1 = -1, nd = 0;
  for (int ch = 0; ch < 256; ch++) {
    if (!cnt[ch]) continue;
    nd++;
    if (cnt[ch] > b0 || (cnt[ch] == b0 && ch > k0)) { b1 = b0; k1 = k0; b0 = cnt[ch]; k0 = ch; }
    else if (cnt[ch] > b1 || (cnt[ch] == b1 && ch > k1)) { b1 = cnt[ch]; k1 = ch; }
  }
  (void)k1;
  *c0 = b0; *ch0 = (uint8_t)k0; *c1 = b1; *ndist = nd;
}

static const uint8_t* novel_from_reads(Arena* ar, const OLead* best, OLead** others, int64_t n_others, int klen, int skip) {
  const double minspan = 0.2, minalns = 0.25, minident = 0.5;
  const int consensus_min = 2, minident_abs = 5, minbestdiff = 3;
  int maxshift = klen;
  int64_t L = best->seq_len;
  const uint8_t* B = best->seq;
  /* anchors: k-mers seen exactly once among sampled positions */
  int64_t npos = 0; for (int64_t i = 0; i < L - klen; i += skip) npos++;
  int64_t hs = 16; while (hs < 2 * npos + 2) hs <<= 1;
  KSlot* tab = (KSlot*)arena_alloc(ar, (size_t)hs * sizeof(KSlot));
  memset(tab, 0, (size_t)hs * sizeof(KSlot));
  for (int64_t i = 0; i < L - klen; i += skip) {
    uint64_t key = kmer_key(B + i, klen);
    int64_t p = (int64_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (hs - 1);
    while (tab[p].state && tab[p].key != key) p = (p + 1) & (hs - 1);
    if (tab[p].state == 0) { tab[p].state = 1; tab[p].key = key; tab[p].pos = (int32_t)i; }
    else tab[p].state = 2;
  }
  uint8_t** alignments = (uint8_t**)arena_alloc(ar, (size_t)(n_others + 1) * sizeof(uint8_t*));
  int64_t nal = 0;
  uint8_t* conseq = (uint8_t*)arena_alloc(ar, (size_t)L + 16);
  for (int64_t r = 0; r < n_others; r++) {
    const OLead* ld = others[r];
    const uint8_t* S = ld->seq; int64_t SL = ld->seq_len;
    int have_last = 0; int64_t last_i = 0, last_j = 0, clen = 0, span = 0;
    for (int64_t j = 0; j < SL - klen; j += skip) {
      uint64_t key = kmer_key(S + j, klen);
      int64_t p = (int64_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (hs - 1);
      while (tab[p].state && tab[p].key != key) p = (p + 1) & (hs - 1);
      if (tab[p].state != 1) continue;
      int64_t i = tab[p].pos;
      if (iabs64(i - j) > maxshift) continue;
      if (have_last && i <= last_i) continue;
      if (!have_last) {
        if (j > 0) { memset(conseq, '-', (size_t)i); clen = i; }
      } else {
        int64_t fwd_i = i - last_i, fwd_j = j - last_j;
        if (clen + fwd_j > L) fwd_j = L - clen;
        if (fwd_i == fwd_j && fwd_j > 0) {
          span += (j - last_j);
          int64_t m = 0;
          for (int64_t l = 1; l <= j - last_j; l++) if (S[last_j + l] == B[last_i + l]) m++;
          double ident = (double)m / (double)(j - last_j);
          if (ident >= minident) memcpy(conseq + clen, S + last_j, (size_t)fwd_j);
          else memset(conseq + clen, '-', (size_t)fwd_j);
          clen += fwd_j;
        } else {
          if (fwd_j > 0) { memset(conseq + clen, '-', (size_t)fwd_j); clen += fwd_j; }
        }
      }
      have_last = 1; last_i = i; last_j = j;
    }
    if (clen < L) { memset(conseq + clen, '-', (size_t)(L - clen)); clen = L; }
    uint8_t* cn = (uint8_t*)arena_alloc(ar, (size_t)L + 1);
    int64_t h = 0;
    while (h < L) {
      if (conseq[h] == '-') { cn[h] = '-'; h++; }
      else {
        int64_t h0 = h, ident = 0;
        while (h < L && conseq[h] != '-') { ident += (B[h] == conseq[h]); h++; }
        int64_t bl = h - h0;
        if ((double)ident / (double)bl > minident && ident > minident_abs) memcpy(cn + h0, conseq + h0, (size_t)bl);
        else memset(cn + h0, '-', (size_t)bl);
      }
    }
    if ((double)span / (double)L > minspan) alignments[nal++] = cn;
  }
  double maxal = (double)(1 + nal); /* the "^_" test never excludes '-' (consensus.py:365-368) */
  if (L == 0) maxal = 1.0;
  uint8_t* flat = (uint8_t*)arena_alloc(ar, (size_t)L + 1);
  uint8_t* col = (uint8_t*)arena_alloc(ar, (size_t)nal + 2);
  for (int64_t i = 0; i < L; i++) {
    int64_t k = 0;
    col[k++] = B[i];
    for (int64_t a = 0; a < nal; a++) if (alignments[a][i] != '-') col[k++] = alignments[a][i];
    int64_t nvotes = k - 1;
    if (nvotes < consensus_min || (double)nvotes / maxal < minalns) flat[i] = B[i];
    else {
      int64_t c0, c1; uint8_t ch0; int nd;
      most_common2(col, k, &c0, &ch0, &c1, &nd);
      if (nd > 1 && c0 - c1 >= minbestdiff) flat[i] = ch0; else flat[i] = B[i];
    }
  }
  return flat;
}

/* annotate_sv INS branch (postprocessing.py:33-66) */
static void ins_consensus(Task* T, OCall* oc) {
  Arena* ar = T->ar; const snf_config_t* cfg = T->cfg;
  OCluster* cl = oc->cluster; snf_call_t* c = &oc->c;
  OLead** m = (OLead**)arena_alloc(ar, (size_t)(cl->leads.n + 1) * sizeof(OLead*));
  int64_t k = 0;
  for (int64_t i = 0; i < cl->leads.n; i++) if (cl->leads.a[i]->seq != NULL) m[k++] = cl->leads.a[i];
  if (k == 0) return;
  int64_t best = 0;
  double best_diff = (double)iabs64(m[0]->seq_len - c->svlen) + (double)iabs64((int64_t)m[0]->ref_start - c->pos) * 1.5;
  for (int64_t i = 1; i < k; i++) {
    double d = (double)iabs64(m[i]->seq_len - c->svlen) + (double)iabs64((int64_t)m[i]->ref_start - c->pos) * 1.5;
    if (d < best_diff) { best = i; best_diff = d; }
  }
  OLead* bl = m[best];
  for (int64_t i = best; i + 1 < k; i++) m[i] = m[i + 1];
  k--;
  if (k >= cfg->consensus_min_reads && !cfg->no_consensus) {
    int skip = cfg->consensus_kmer_skip_base + (int)((double)bl->seq_len * cfg->consensus_kmer_skip_seqlen_mult);
    oc->alt = novel_from_reads(ar, bl, m, k, cfg->consensus_kmer_len, skip);
  } else oc->alt = bl->seq;
  oc->alt_len = bl->seq_len;
}

/* qc_sv_post_annotate (postprocessing.py:444-600) */
static int qc_sv_post_annotate(OCall* oc, const snf_config_t* cfg, double qc_nm_threshold_task, double cov_avg_total) {
  snf_call_t* c = &oc->c; OCluster* cl = oc->cluster;
  int t = c->svtype;
  double af = isnan(c->vaf) ? 0.0 : c->vaf;
  int sv_is_mosaic = af <= cfg->mosaic_af_max;
  int gt_dot = 0; /* genotype alleles are never "." on this path */
  if ((c->cov[2] < cfg->qc_coverage && (!c->gt_set || (!gt_dot && c->gt_a + c->gt_b < 2))) &&
      (t != SNF_DEL && iabs64(c->svlen) > cfg->long_del_length)) { c->filter = SNF_F_COV_MIN_GT; return 0; }
  if (cfg->mosaic && !sv_is_mosaic) { if (!qc_sv_support(c, cov_avg_total, cfg)) return 0; }
  int qc_nm = cfg->qc_nm;
  double thr = qc_nm_threshold_task * cfg->qc_nm_mult;
  if (cfg->mosaic && sv_is_mosaic) qc_nm = cfg->mosaic_qc_nm;
  if (qc_nm && c->nm > thr && (!c->gt_set || c->gt_b == 0)) { c->filter = SNF_F_ALN_NM; return 0; }
  if (!cfg->mosaic && sv_is_mosaic) {
    int skip_this_dup = (t == SNF_DUP) && af >= cfg->dev_min_dup_vaf;
    if (!skip_this_dup) { c->filter = SNF_F_MOSAIC_VAF; return 0; }
  }
  if (cfg->mosaic && sv_is_mosaic) {
    int min_mosaic_support = cfg->mosaic_min_reads;
    int accepted = (t == SNF_INS || t == SNF_DEL || t == SNF_DUP || t == SNF_INV || t == SNF_BND);
    if (!isnan(c->stdev_len) && accepted) {
      int filter_low_supp = ((!c->precise || c->stdev_len / (double)iabs64(c->svlen) > 0.1 || c->stdev_pos > 5) &&
                             1 <= cfg->max_svlen_mosaic);
      min_mosaic_support = (t == SNF_BND || t == SNF_INV || filter_low_supp) ? cfg->mosaic_min_reads : cfg->mosaic_min_reads - 1;
    }
    if (c->support < min_mosaic_support) { c->filter = SNF_F_SUPPORT_MIN; return 0; }
    if (t != SNF_BND && iabs64(c->svlen) > cfg->max_svlen_mosaic) { c->filter = SNF_F_SVLEN_MAX_MOSAIC; return 0; }
  }
  if (t != SNF_BND) {
    int is_long_ins = (t == SNF_INS && c->svlen >= cfg->long_ins_length);
    if (!(cfg->mosaic && sv_is_mosaic) && cfg->qc_strand) {
      if (!is_long_ins && distinct_strands(cl) < 2) { c->filter = SNF_F_STRAND; return 0; }
    } else if ((cfg->mosaic && sv_is_mosaic) && cfg->mosaic_qc_strand) {
      if (!is_long_ins && distinct_strands(cl) < 2 && c->support >= cfg->mosaic_use_strand_thresholds) { c->filter = SNF_F_STRAND_MOSAIC; return 0; }
    }
  }
  if (cfg->mosaic && sv_is_mosaic) {
    if ((t == SNF_INV || t == SNF_DUP) && c->svlen < cfg->mosaic_qc_invdup_min_length) { c->filter = SNF_F_SVLEN_MIN_MOSAIC; return 0; }
  }
  if (c->cov[2] < cfg->qc_coverage && t != SNF_DEL && t != SNF_INS) {
    int64_t lhs = (t == SNF_INV) ? c->svlen : 0; /* (svtype == "INV" and svlen) > long_inv_length */
    if (lhs > cfg->long_inv_length && !(cfg->mosaic && sv_is_mosaic)) { /* pass */ }
    else { c->filter = SNF_F_COV_MIN; return 0; }
  }
  if (cfg->mosaic) {
    if (sv_is_mosaic && (af < cfg->mosaic_af_min || af > cfg->mosaic_af_max)) { c->filter = SNF_F_MOSAIC_VAF; return 0; }
    else if (!sv_is_mosaic && !cfg->mosaic_include_germline) { c->filter = SNF_F_NOT_MOSAIC_VAF; return 0; }
    if (sv_is_mosaic && t != SNF_BND && t != SNF_SINGLE_LEFT && t != SNF_SINGLE_RIGHT) {
      int64_t close = 0;
      for (int64_t i = 0; i < cl->leads.n; i++) {
        OLead* l = cl->leads.a[i];
        if (l->qry_start <= cfg->dev_min_close_edge_dist || iabs64((int64_t)l->read_len - l->qry_start) <= cfg->dev_min_close_edge_dist) close++;
      }
      if ((double)close / (double)c->support >= cfg->dev_min_read_close_edge_prop) { c->filter = SNF_F_MOSAIC_SV_CLOSE_EDGE; return 0; }
    }
  }
  return 1;
}

/* Task.rescue_phasing (parallel.py:203-249) */
static void rescue_phasing(Task* T, OCall* oc) {
  const snf_config_t* cfg = T->cfg; snf_call_t* c = &oc->c; OCluster* cl = oc->cluster;
  if (!cfg->mode_call_sample) return;
  int64_t n = cl->leads.n;
  double* v = (double*)arena_alloc(T->ar, (size_t)(n + 1) * sizeof(double));
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; i++) { double x = cl->leads.a[i]->nm; if (isnan(x)) v[i] = 0; else { v[i] = x; cnt++; } }
  double sv_nm = np_pairwise_sum(v, n) / (double)cnt; /* np.nanmean */
  if (sv_nm > cfg->genotype_error || n <= 3) return;
  if (!c->ph_set) return;
  if (!c->ph_hp_pass) return;
  int hp = c->ph_hp;
  int32_t all_reads, sv_reads;
  if (hp == 1) { all_reads = cl->hap[4]; sv_reads = cl->hap[1]; }
  else if (hp == 2) { all_reads = cl->hap[5]; sv_reads = cl->hap[2]; }
  else return;
  if (all_reads == 0) return;
  if ((double)sv_reads / (double)all_reads >= 0.75) {
    if (c->filter == SNF_F_MOSAIC_VAF) { c->filter = SNF_F_PASS; c->gt_b = 1; c->qc = 1; }
  }
}

/* ------------------------------------------------------------------ driver */
static void build_coverage(Task* T) {
  const snf_task_input_t* in = T->in;
  int64_t L = in->contig_len;
  int32_t* diff = (int32_t*)calloc((size_t)L + 2, sizeof(int32_t));
  int64_t nb = L / T->cfg->cluster_binsize + 2;
  T->nbins = nb;
  int32_t* hd[3];
  for (int h = 0; h < 3; h++) hd[h] = (int32_t*)calloc((size_t)nb + 2, sizeof(int32_t));
  int bs = T->cfg->cluster_binsize;
  for (int64_t r = 0; r < in->n_reads; r++) {
    int64_t s = in->read_start[r], e = in->read_end[r];
    int64_t cs = s < 0 ? 0 : (s > L ? L : s), ce = e < 0 ? 0 : (e > L ? L : e);
    if (ce > cs) { diff[cs]++; diff[ce]--; }
    /* record_hap_ref(hp, int(s/bs)*bs, int(e/bs)*bs, bs): range(pos, end, step) */
    if (s >= 0 && s < L) {
      int64_t b0 = s / bs, b1 = e / bs;
      if (b1 > nb) b1 = nb;
      if (b1 > b0) { hd[in->read_hp[r]][b0]++; hd[in->read_hp[r]][b1]--; }
    }
  }
  T->coverage = (uint16_t*)malloc((size_t)(L + 1) * sizeof(uint16_t));
  int64_t run = 0; uint64_t total = 0;
  for (int64_t i = 0; i < L; i++) { run += diff[i]; T->coverage[i] = (uint16_t)run; }
  free(diff);
  /* LeadProvider._mask_N_coverage (leadprov.py:420-443): coverage[mask == 78] = 0, after all regions have been read */
  for (int64_t k = 0; k < in->n_nmask; k++)
    for (int64_t i = in->nmask_start[k]; i < in->nmask_end[k] && i < L; i++) T->coverage[i] = 0;
  for (int64_t i = 0; i < L; i++) total += T->coverage[i];
  T->coverage_average_total = L > 0 ? (double)total / (double)L : NAN;
  for (int h = 0; h < 3; h++) {
    T->hapref[h] = (uint16_t*)malloc((size_t)(nb + 1) * sizeof(uint16_t));
    int64_t r2 = 0;
    for (int64_t b = 0; b < nb; b++) { r2 += hd[h][b]; T->hapref[h][b] = (uint16_t)(r2 > 65535 ? 65535 : r2); }
    free(hd[h]);
  }
}

static double g_hot_seconds = 0.0;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
/* seconds spent inside call_candidates + finalize_candidates of the last snf_oracle_run (excludes building the
 * dense coverage vector / hap tables, which the reference does during BAM extraction, leadprov.py:474-578) */
double snf_oracle_hot_seconds(void) { return g_hot_seconds; }

static void run_task(Task* T, int do_finalize) {
  const snf_task_input_t* in = T->in; const snf_config_t* cfg = T->cfg; Arena* ar = T->ar;
  T->n = in->n_leads;
  T->leads = (OLead*)arena_alloc(ar, (size_t)(T->n + 1) * sizeof(OLead));
  for (int64_t i = 0; i < T->n; i++) {
    OLead* l = &T->leads[i];
    l->ref_start = in->ref_start[i]; l->ref_end = in->ref_end[i]; l->qry_start = in->qry_start[i]; l->qry_end = in->qry_end[i];
    l->has_svlen = in->svlen[i] != SNF_SVLEN_NONE; l->svlen = l->has_svlen ? in->svlen[i] : 0;
    l->read_len = in->read_len[i]; l->qname = in->qname_id[i]; l->read_id = in->read_id[i]; l->ps = in->ps_rank[i];
    l->mate_contig = in->mate_contig[i]; l->mate_ref_start = in->mate_ref_start[i];
    l->svtype = in->svtype[i]; l->strand = in->strand[i]; l->mapq = in->mapq[i]; l->source = in->source[i];
    l->hap = in->hap[i]; l->is_sa = in->is_sa[i]; l->is_first = in->bnd_is_first[i]; l->is_reverse = in->bnd_is_reverse[i];
    l->nm = in->nm[i];
    if (in->seq_len[i] >= 0) { l->seq = in->seq_pool + in->seq_off[i]; l->seq_len = in->seq_len[i]; } else { l->seq = NULL; l->seq_len = 0; }
    /* build_leadtab keeps only leads inside the task region (leadprov.py:464-468) */
    l->orig = (l->ref_start >= 0 && l->ref_start < in->contig_len) ? i : -1;
  }
  build_coverage(T);
  double t_hot0 = now_s();
  T->sv_id = in->sv_id_start;
  /* Task.call_candidates (parallel.py:104-127) */
  for (int svtype = 0; svtype < SNF_NTYPES; svtype++) {
    ClVec cls = {0};
    resolve(T, svtype, &cls);
    for (int64_t i = 0; i < cls.n; i++) {
      OCluster* cl = cls.a[i];
      /* Cluster.get_sa_count (cluster.py:79-82) */
      int64_t sa = 0, all = cl->leads.n + (cl->has_long ? cl->leads_long.n : 0);
      for (int64_t t = 0; t < cl->leads.n; t++) sa += cl->leads.a[t]->is_sa;
      if (cl->has_long) for (int64_t t = 0; t < cl->leads_long.n; t++) sa += cl->leads_long.a[t]->is_sa;
      cl->sa_count = (int32_t)sa; cl->sa_frac = (double)sa / (double)all;
      call_from(T, cl, svtype);
    }
  }
  annotate_coverage(T);
  if (T->status != SNF_TASK_OK) { T->calls.n = 0; g_hot_seconds += now_s() - t_hot0; return; }
  if (!do_finalize) { g_hot_seconds += now_s() - t_hot0; return; }
  /* Task.finalize_candidates (parallel.py:129-201) */
  for (int64_t i = 0; i < T->calls.n; i++) {
    OCall* oc = &T->calls.a[i]; snf_call_t* c = &oc->c;
    c->qc = c->qc && qc_sv(oc, cfg);
    if (!cfg->mosaic && c->qc) c->qc = c->qc && qc_sv_support(c, T->coverage_average_total, cfg);
    int hp_ret = -1, ps_ret = -1;
    if (cfg->phase) phase_sv(T, oc, &hp_ret, &ps_ret);
    genotype_sv(oc, cfg, hp_ret, ps_ret);
    if (c->svtype == SNF_INS && !cfg->symbolic) ins_consensus(T, oc);
    c->qc = c->qc && qc_sv_post_annotate(oc, cfg, in->qc_nm_threshold, T->coverage_average_total);
    int phasing_rescue = (c->svtype != SNF_BND && iabs64(c->svlen) <= cfg->dev_maxsvlen_extra &&
                          c->support >= (int)((double)cfg->dev_minreads_extra * 0.60));
    if (cfg->phase && !c->qc && phasing_rescue) rescue_phasing(T, oc);
  }
  g_hot_seconds += now_s() - t_hot0;
}

/* ------------------------------------------------------------------ public (tests / bench only) */
typedef struct snf_oracle_result {
  snf_result_t r;
  snf_call_t* calls; uint8_t* alt_pool; uint32_t* rnames; int32_t* status; int64_t* off; double* cov;
} snf_oracle_result_t;

int snf_oracle_run(const snf_config_t* cfg, const snf_task_input_t* tasks, int n_tasks, int do_finalize,
                   snf_oracle_result_t** out) {
  snf_oracle_result_t* R = (snf_oracle_result_t*)calloc(1, sizeof(*R));
  Task* T = (Task*)calloc((size_t)n_tasks, sizeof(Task));
  Arena* arenas = (Arena*)calloc((size_t)n_tasks, sizeof(Arena));
  int64_t ncalls = 0, altlen = 0, rnlen = 0;
  g_hot_seconds = 0.0;
  for (int t = 0; t < n_tasks; t++) {
    T[t].ar = &arenas[t]; T[t].cfg = cfg; T[t].in = &tasks[t]; T[t].task_index = t;
    run_task(&T[t], do_finalize);
    ncalls += T[t].calls.n;
    for (int64_t i = 0; i < T[t].calls.n; i++) { if (T[t].calls.a[i].alt) altlen += T[t].calls.a[i].alt_len; rnlen += T[t].calls.a[i].rn_n; }
    free(T[t].coverage); T[t].coverage = NULL;
    for (int h = 0; h < 3; h++) { free(T[t].hapref[h]); T[t].hapref[h] = NULL; }
  }
  R->calls = (snf_call_t*)malloc((size_t)(ncalls + 1) * sizeof(snf_call_t));
  R->alt_pool = (uint8_t*)malloc((size_t)altlen + 1);
  R->rnames = (uint32_t*)malloc((size_t)(rnlen + 1) * sizeof(uint32_t));
  R->status = (int32_t*)malloc((size_t)(n_tasks + 1) * sizeof(int32_t));
  R->off = (int64_t*)malloc((size_t)(n_tasks + 2) * sizeof(int64_t));
  R->cov = (double*)malloc((size_t)(n_tasks + 1) * sizeof(double));
  int64_t k = 0, ao = 0, ro = 0;
  for (int t = 0; t < n_tasks; t++) {
    R->off[t] = k; R->status[t] = T[t].status; R->cov[t] = T[t].coverage_average_total;
    for (int64_t i = 0; i < T[t].calls.n; i++) {
      OCall* oc = &T[t].calls.a[i];
      snf_call_t c = oc->c;
      if (oc->alt) { c.alt_off = ao; c.alt_len = (int32_t)oc->alt_len; memcpy(R->alt_pool + ao, oc->alt, (size_t)oc->alt_len); ao += oc->alt_len; }
      else { c.alt_off = 0; c.alt_len = -1; }
      c.rn_off = ro; c.rn_len = (int32_t)oc->rn_n;
      if (oc->rn_n) memcpy(R->rnames + ro, oc->rn, (size_t)oc->rn_n * sizeof(uint32_t));
      ro += oc->rn_n;
      R->calls[k++] = c;
    }
    free(T[t].calls.a);
    arena_free(&arenas[t]);
  }
  R->off[n_tasks] = k;
  R->r.n_calls = k; R->r.calls = R->calls; R->r.alt_pool_len = ao; R->r.alt_pool = R->alt_pool;
  R->r.rnames_len = ro; R->r.rnames = R->rnames; R->r.n_tasks = n_tasks; R->r.task_status = R->status;
  R->r.task_call_off = R->off; R->r.coverage_average_total = R->cov;
  free(T); free(arenas);
  *out = R;
  return 0;
}

const snf_result_t* snf_oracle_result(const snf_oracle_result_t* R) { return &R->r; }

void snf_oracle_free(snf_oracle_result_t* R) {
  if (!R) return;
  free(R->calls); free(R->alt_pool); free(R->rnames); free(R->status); free(R->off); free(R->cov); free(R);
}

/* exact global (NW) unit-cost edit distance, the definition of edlib.align(a,b)["editDistance"]
 * with default arguments (sv.py:287, snfp.py:103); plain two-row DP */
int64_t snf_oracle_edit_distance(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) {
  if (la < lb) { const uint8_t* t = a; a = b; b = t; int64_t tl = la; la = lb; lb = tl; }
  int64_t* prev = (int64_t*)malloc((size_t)(lb + 1) * sizeof(int64_t));
  int64_t* cur = (int64_t*)malloc((size_t)(lb + 1) * sizeof(int64_t));
  for (int64_t j = 0; j <= lb; j++) prev[j] = j;
  for (int64_t i = 1; i <= la; i++) {
    cur[0] = i;
    for (int64_t j = 1; j <= lb; j++) {
      int64_t v = prev[j - 1] + (a[i - 1] != b[j - 1]);
      if (prev[j] + 1 < v) v = prev[j] + 1;
      if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
      cur[j] = v;
    }
    int64_t* t = prev; prev = cur; cur = t;
  }
  int64_t d = prev[lb];
  free(prev); free(cur);
  return d;
}

/* The same distance by the bit-parallel algorithm edlib implements (Myers 1999 in Hyyro's formulation for the GLOBAL distance, 64 rows of the DP
 * matrix per machine word, blocks chained through the horizontal carries): what a CPU run of the reference spends its alignment time in when
 * edlib is present.  Used as the edlib STAND-IN of bench.py --config 4's reference baseline (edlib itself is absent from this image); pinned to
 * snf_oracle_edit_distance by tests/test_edit_distance.py.  O(ceil(m / 64) * n) word operations. */
int64_t snf_oracle_edit_distance_myers(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) {
  /* pattern = the shorter string (rows), text = the longer one (columns) */
  if (la > lb) { const uint8_t* t = a; a = b; b = t; int64_t tl = la; la = lb; lb = tl; }
  const int64_t m = la, n = lb;
  if (m == 0) return n;
  const int64_t W = (m + 63) / 64;
  uint64_t* peq = (uint64_t*)calloc((size_t)(256 * W), sizeof(uint64_t));
  uint64_t* Pv = (uint64_t*)malloc((size_t)W * sizeof(uint64_t));
  uint64_t* Mv = (uint64_t*)calloc((size_t)W, sizeof(uint64_t));
  for (int64_t i = 0; i < m; i++) peq[(size_t)a[i] * (size_t)W + (size_t)(i >> 6)] |= 1ull << (i & 63);
  for (int64_t w = 0; w < W; w++) Pv[w] = ~0ull;
  const int last_bits = (int)(m - 64 * (W - 1));            /* rows in the last word: 1..64 */
  const uint64_t last_mask = 1ull << (last_bits - 1);
  int64_t score = m;
  for (int64_t j = 0; j < n; j++) {
    const uint64_t* eqc = peq + (size_t)b[j] * (size_t)W;
    int hin = 1;                                            /* global alignment: the top row of the matrix is 0, 1, 2, ... (+1 per column) */
    for (int64_t w = 0; w < W; w++) {
      uint64_t Eq = eqc[w];
      const uint64_t pv = Pv[w], mv = Mv[w];
      const uint64_t hin_neg = hin < 0 ? 1ull : 0ull;
      const uint64_t Xv = Eq | mv;
      Eq |= hin_neg;
      const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
      uint64_t Ph = mv | ~(Xh | pv);
      uint64_t Mh = pv & Xh;
      int hout = 0;
      const uint64_t top = (w == W - 1) ? last_mask : (1ull << 63);
      if (Ph & top) hout = 1; else if (Mh & top) hout = -1;
      Ph <<= 1; Mh <<= 1;
      if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
      Pv[w] = Mh | ~(Xv | Ph);
      Mv[w] = Ph & Xv;
      hin = hout;
    }
    score += hin;                                           /* hin is now the horizontal delta of the last row */
  }
  free(peq); free(Pv); free(Mv);
  return score;
}

/* ------------------------------------------------------------------ combine (multi-sample) -- test infrastructure
 * cluster.resolve_block_groups (cluster.py:356-390) + SVGroup.from_candidate / align_call / add_candidate
 * (sv.py:263-318), followed literally; edlib.align(a,b)["editDistance"] = snf_oracle_edit_distance (exact DP). */
typedef struct OGroup {
  double pos_mean, len_mean, mate_mean; int is_int_mate;
  int32_t size, mate_contig;
  const uint8_t* alt; int64_t alt_len;
  uint8_t* included; /* [n_sample_ids] */
} OGroup;

int snf_oracle_combine_resolve(const snf_config_t* cfg, const snf_combine_problem_t* q) {
  int nc = q->n_cands, ng = q->n_groups, ns = q->n_sample_ids > 0 ? q->n_sample_ids : 1;
  OGroup* G = (OGroup*)calloc((size_t)(nc + ng + 1), sizeof(OGroup));
  for (int g = 0; g < ng; g++) {
    G[g].pos_mean = q->g_pos_mean[g]; G[g].len_mean = q->g_len_mean[g]; G[g].mate_mean = q->g_mate_mean ? q->g_mate_mean[g] : 0;
    G[g].size = q->g_size[g]; G[g].mate_contig = q->g_mate_contig ? q->g_mate_contig[g] : 0;
    G[g].alt = q->g_alt_pool + q->g_alt_off[g]; G[g].alt_len = q->g_alt_off[g + 1] - q->g_alt_off[g];
    G[g].included = (uint8_t*)calloc((size_t)ns, 1);
    for (int64_t k = q->g_samples_off[g]; k < q->g_samples_off[g + 1]; k++) G[g].included[q->g_samples[k]] = 1;
  }
  /* sorted(svcands, key=lambda cand: cand.support, reverse=True): stable */
  SortKey* sk = (SortKey*)malloc((size_t)(nc + 1) * sizeof(SortKey));
  for (int i = 0; i < nc; i++) { sk[i].key = -(int64_t)q->support[i]; sk[i].idx = i; }
  qsort(sk, (size_t)nc, sizeof(SortKey), cmp_sortkey);
  for (int oi = 0; oi < nc; oi++) {
    int c = (int)sk[oi].idx;
    int best = -1; double best_dist = INFINITY;
    if (q->svtype == SNF_BND) {
      for (int g = 0; g < ng; g++) {
        double dist = fabs(G[g].pos_mean - (double)q->pos[c]) + fabs(G[g].mate_mean - (double)q->mate_ref_start[c]);
        if (dist < best_dist && dist <= (double)(cfg->cluster_merge_bnd * 2) && G[g].mate_contig == q->mate_contig[c]) {
          if (!cfg->combine_separate_intra || !G[g].included[q->sample_id[c]]) { best = g; best_dist = dist; }
        }
      }
    } else {
      for (int g = 0; g < ng; g++) {
        double alen = fabs((double)q->svlen[c]);
        double dist = fabs(G[g].pos_mean - (double)q->pos[c]) + fabs(fabs(G[g].len_mean) - alen);
        double minlen = fabs(G[g].len_mean) < alen ? fabs(G[g].len_mean) : alen;
        if (minlen > 0 && dist < best_dist && dist <= (double)cfg->combine_match * sqrt(minlen) && dist <= (double)cfg->combine_match_max) {
          if (!cfg->combine_separate_intra || !G[g].included[q->sample_id[c]]) {
            int ok = 1;
            if (cfg->combine_pctseq != 0.0) {
              int64_t d = snf_oracle_edit_distance(G[g].alt, G[g].alt_len, q->alt_pool + q->alt_off[c], q->alt_off[c + 1] - q->alt_off[c]);
              ok = ((G[g].len_mean - (double)d) / G[g].len_mean) > cfg->combine_pctseq;
            }
            if (ok) { best = g; best_dist = dist; }
          }
        }
      }
    }
    if (best < 0) {
      OGroup* g = &G[ng];
      g->pos_mean = (double)q->pos[c]; g->len_mean = fabs((double)q->svlen[c]);
      g->mate_mean = q->mate_ref_start ? (double)q->mate_ref_start[c] : 0; g->size = 1;
      g->mate_contig = q->mate_contig ? q->mate_contig[c] : 0;
      g->alt = q->alt_pool + q->alt_off[c]; g->alt_len = q->alt_off[c + 1] - q->alt_off[c];
      g->included = (uint8_t*)calloc((size_t)ns, 1); g->included[q->sample_id[c]] = 1;
      q->out_group[c] = ng++;
    } else {
      OGroup* g = &G[best];
      double n = (double)g->size;
      g->pos_mean *= n; g->len_mean *= n;
      g->pos_mean += (double)q->pos[c]; g->len_mean += fabs((double)q->svlen[c]);
      if (q->svtype == SNF_BND) { g->mate_mean *= n; g->mate_mean += (double)q->mate_ref_start[c]; }
      g->size++;
      g->pos_mean /= (double)g->size; g->len_mean /= (double)g->size;
      g->included[q->sample_id[c]] = 1;
      if (q->svtype == SNF_BND) g->mate_mean /= (double)g->size;
      q->out_group[c] = best;
    }
  }
  for (int g = 0; g < ng; g++) free(G[g].included);
  free(G); free(sk);
  return 0;
}
