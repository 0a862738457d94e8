// snf_extract.hip - SV-signature extraction from raw BAM alignment records.
//
// Reference behaviour restated on the GPU (one Region of one contig):
//   iter_region       leadprov.py:474-578   read filter, read ids, coverage / REF-haplotype bookkeeping, NM side channel
//   read_iterindels   leadprov.py:580-655   INS / DEL / clip signatures from the CIGAR
//   Lead.for_bnd      leadprov.py:57-132    BND signature from the first SA element
//   CIGAR_analyze     leadprov.py:144-178   clip / span summary of an SA CIGAR string
//   read_itersplits   leadprov.py:221-355   split alignments -> leads
//   classify_splits   sv.py:649-782         INS / DEL / DUP / INV geometry of consecutive splits
//   build_leadtab     leadprov.py:445-472   keep the leads of this contig inside [start, end)
// pysam properties are computed from the record bytes (SAM/BAM spec 4.2; htslib bam_endpos, pysam getQueryStart/End).
//
// Mapping: one WAVE (and one workgroup) per alignment record.  A record is a chain of dependent accesses (offset -> fixed fields ->
// CIGAR ends / tags / CIGAR body -> SA string); what the wave form does about it: the 24 bytes of fixed fields arrive as ONE load
// (a dword per lane, read back into scalar registers); then, in one round trip, the first and last four CIGAR operations (the
// clip scans of getQueryStart / getQueryEnd), the record's whole auxiliary region (16 bytes per lane into LDS: the tag walk and
// the lane-0 SA parser then read LDS, a tag costing one 8-byte read) and the first two steps of the CIGAR.  The CIGAR (4 B per
// operation, thousands of operations per long read) is walked 256 operations per step (one 16-byte load per lane, two steps
// ahead): two DPP wave prefix sums give every lane the read / reference position of its operations, a third ranks the
// signature-bearing operations so leads come out in CIGAR order.
// Inserted sequence is decoded from the 4-bit packed read by the whole wave.  The per-record split-alignment logic (a handful of
// SA elements) is sequential and runs on lane 0 with its segment table in LDS.  Two passes over the records (count, emit)
// around three device scans give exact output offsets, so the leads land in `record_lead` order without atomics.
// HBM-bound byte/integer work: no MFMA.  Algorithmic bytes per record: 36 B core + name + 4 B x n_cigar + aux bytes,
// plus the inserted bases read (0.5 B) and written (1 B); sequence and quality bytes of the read are never touched.
// The thread form (WAVE == false) is what the host emulation executes (tests/emu) and SNF_EXTRACT_THREAD=1 selects.
#include "snf_rt.h"
#include "../../include/sniffles_amd.h"

#include <algorithm>
#include <string>
#include <vector>

#include <rocprim/device/device_scan.hpp>

namespace snf {

enum {
  XE_OK = 0, XE_NOCIGAR, XE_CIGAROP, XE_CLIP, XE_AUX, XE_TAGTYPE, XE_HP, XE_NOSEQ, XE_SA_FIELDS, XE_SA_NUMBER,
  XE_SA_STRAND, XE_SA_CONTIG, XE_SPLITS, XE_RANGE
};
static const char* const XE_TEXT[] = {
  "ok", "mapped record without CIGAR (reference_end is None)", "CIGAR operation > 8", "invalid clipping in CIGAR (pysam ValueError)",
  "malformed auxiliary field", "NM/HP/PS not an integer tag or SA not a Z tag", "HP tag outside 0..2 (IndexError in record_lead)",
  "record without sequence but an inserted sequence is needed (TypeError)", "SA element without 6 fields (ValueError)",
  "SA number field not a plain integer", "SA strand not '+' or '-'", "SA contig not in the header",
  "more split alignments than the device table holds (64)", "value outside the 32-bit / 8-bit column range"};

#define XMAXSEG 64

struct RecSum {          // per record, written by the counting pass (64 bytes)
  int64_t seq_bytes;     // bytes this record adds to the sequence pool
  int64_t ps;            // PS tag (has_ps)
  double nm;             // (NM - large indels) / (query_alignment_length + 1), or -1
  int32_t sa_rel, sa_len;     // the SA string: byte offset inside the record's auxiliary region (-1: no SA tag), length without its NUL
  int32_t n_leads, ref_end, nm_tag;
  int32_t walk_lo, walk_hi;   // CIGAR steps [walk_lo, walk_hi) hold every lead-bearing operation (emit pass walks only these)
  uint32_t walk_pr, walk_pf;  // read / reference position at walk_lo
  uint8_t accept, has_nm, has_ps, hp;
};
static_assert(sizeof(RecSum) == 64, "one 64-byte record per alignment record");

struct Seg {             // one (split) alignment of a read in classify_splits
  int32_t contig, rs, re, qs, qe, mapq;
  int32_t hint_type, hint_start, hint_len;   // hint_type -1: none; hint_len SNF_SVLEN_NONE: None
  int32_t seq_a, seq_n;                      // slice of the read sequence attached to the split (seq_n -1: None)
  int64_t job_dst;                           // emit pass: pool offset the slice is copied to, -1: nothing to copy
  uint8_t rev, source;
};

struct ExView {
  snf_extract_config_t cfg;
  const uint8_t* blob; const int64_t* rec_off; const uint32_t* qname_rank; int64_t n_records;
  const uint32_t* order;     // wave form: the record workgroup w takes - the records of the region's contig by falling CIGAR length (built at upload)
  int32_t region_ref_id, region_rank, region_start, region_end; uint32_t read_id_offset;
  const uint64_t* ctg_hash; const int32_t* ctg_rank; int32_t n_contigs;
  RecSum* sum;
  unsigned long long* err;   // min over (record << 8 | code); ~0: none
  // scan inputs / outputs (n_records + 1)
  int64_t *c_acc, *c_leads, *c_seq;
  double* c_nm;                       // per record: its NM ratio if it counts for average_regional_nm (accepted, NM tag, advanced tags), else +0.0
  int64_t* c_ps; uint8_t* c_psf;      // per record: its PS tag and whether it counts (accepted, tag present) - what the host ranks
  unsigned long long* tot;            // [0] first error (= err), [1..3] reads, leads, sequence bytes, [4] NM summands: one copy to the host
  const int64_t *read_idx, *lead_off, *seq_off;
  const int64_t* ps_val; const int32_t* ps_rank; int32_t n_ps, ps_null_rank;
  // lead columns
  int32_t *o_ref_start, *o_ref_end, *o_qry_start, *o_qry_end, *o_svlen, *o_read_len, *o_ps, *o_mate_contig, *o_mate_pos, *o_seq_len;
  uint32_t *o_qname, *o_read_id; int64_t* o_seq_off; double* o_nm;
  uint8_t *o_svtype, *o_strand, *o_mapq, *o_source, *o_hap, *o_is_sa, *o_first, *o_rev;
  uint8_t* o_pool;
  int32_t *o_rstart, *o_rend; uint8_t* o_rhp;
  double* nm_out;   // [0] sum, [1] count
#ifdef SNF_XTRACE
  uint32_t* xtrace;   // instrumented build (tools/xtrace.sh): per record {start, total, tags + clips, SA section} in wall-clock ticks (10 ns)
  int xtrace_emit;    // which pass leaves its stamps
#endif
};
#ifdef SNF_XTRACE
#define XT_DECL unsigned long long xt_[4] = {(unsigned long long)wall_clock64(), 0, 0, 0};
#define XT(k) xt_[k] = wall_clock64();
#define XT_ADD(k, t0) xt_[k] += wall_clock64() - (t0);
#define XT_FLUSH() do { if (v.xtrace && lane == 0 && (int)EMIT == v.xtrace_emit) { uint32_t* o_ = v.xtrace + 4 * rec; o_[0] = (uint32_t)xt_[0]; \
    o_[1] = (uint32_t)(wall_clock64() - xt_[0]); o_[2] = (uint32_t)(xt_[1] ? xt_[1] - xt_[0] : 0); o_[3] = (uint32_t)xt_[2]; } } while (0)
#else
#define XT_DECL
#define XT(k)
#define XT_ADD(k, t0)
#define XT_FLUSH()
#endif

// ---- lane helpers: WAVE == true only exists in device code ----------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define XDEV 1
#else
#define XDEV 0
#endif

template <bool WAVE> SNF_HD int x_lane() {
#if XDEV
  return WAVE ? (int)(threadIdx.x & 63) : 0;
#else
  return 0;
#endif
}
template <bool WAVE> SNF_HD uint64_t x_ballot(bool p) {
#if XDEV
  if (WAVE) return __ballot(p);
#endif
  return p ? 1ull : 0ull;
}
// inclusive prefix sum over the wave: Hillis-Steele inside each row of 16 lanes through DPP row shifts, then the row
// totals are passed on with row_bcast:15 / row_bcast:31 (gfx9 DPP).  Needs all 64 lanes active.
template <bool WAVE> SNF_HD uint32_t x_incl_scan(uint32_t x, int lane) {
#if XDEV
  if (WAVE) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  }
#endif
  (void)lane;
  return x;
}
template <bool WAVE> SNF_HD uint32_t x_bcast(uint32_t x, int src) {
#if XDEV
  if (WAVE) return (uint32_t)__shfl((int)x, src, 64);
#endif
  (void)src;
  return x;
}
template <bool WAVE> SNF_HD int64_t x_bcast64(int64_t x, int src) {
#if XDEV
  if (WAVE) return (int64_t)__shfl((long long)x, src, 64);
#endif
  (void)src;
  return x;
}
template <bool WAVE> SNF_HD void x_wave_sync() {
#if XDEV
  if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#endif
}
// a value every lane holds alike, moved to a scalar register (v_readfirstlane)
template <bool WAVE> SNF_HD uint32_t x_uni(uint32_t x) {
#if XDEV
  if (WAVE) return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
#endif
  return x;
}
template <bool WAVE> SNF_HD uint64_t x_uni64(uint64_t x) { return ((uint64_t)x_uni<WAVE>((uint32_t)(x >> 32)) << 32) | x_uni<WAVE>((uint32_t)x); }
SNF_HD int x_popc(uint64_t m) { return __builtin_popcountll(m); }
SNF_HD int x_ctz(uint64_t m) { return __builtin_ctzll(m); }

SNF_HD uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
// unaligned little-endian dword: two aligned loads and a funnel shift on the device (the blob is padded by 8 bytes)
SNF_HD uint32_t ld_u32(const uint8_t* p) {
#if XDEV
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  const unsigned sh = (unsigned)(a & 3) * 8;
  const uint32_t lo = q[0];
  if (!sh) return lo;
  return (lo >> sh) | (q[1] << (32 - sh));
#else
  uint32_t v; memcpy(&v, p, 4); return v;
#endif
}
// Loads at any byte address of GLOBAL memory (records are byte-aligned in the blob; the device serves misaligned global accesses
// in hardware, so one instruction fetches what two aligned loads and a shift did).  The blob is padded by 16 bytes.
typedef uint32_t __attribute__((aligned(1))) x_u32_any;
typedef uint4 __attribute__((aligned(1))) x_u128_any;
// 8 bytes at any address of the blob OR of the LDS copy of a record's auxiliary region: the two aligned words around the
// address and a funnel shift (an unaligned 8-byte LDS read is served lane by lane); both have >= 16 bytes of slack behind them
SNF_HD uint64_t x_ld8(const uint8_t* p) {
#if XDEV
  const uintptr_t a = (uintptr_t)p;
  const uint64_t* q = (const uint64_t*)(a & ~(uintptr_t)7);
  const unsigned sh = (unsigned)(a & 7) * 8;
  const uint64_t lo = q[0], hi = q[1];
  return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
#else
  uint64_t w; memcpy(&w, p, 8); return w;
#endif
}
// the first six dwords of a record (block_size, refID, pos, l_read_name | mapq | bin, n_cigar_op | flag, l_seq).  Wave form: one
// dword per lane, read back with v_readlane - the values sit in scalar registers from then on (everything derived from them is
// wave-uniform and costs no vector registers)
template <bool WAVE> SNF_HD void x_header(const uint8_t* R, int lane, uint32_t* hd) {
#if XDEV
  if (WAVE) {
    uint32_t w = 0;
    if (lane < 6) w = *(const x_u32_any*)(R + 4 * lane);
#pragma unroll
    for (int k = 0; k < 6; k++) hd[k] = (uint32_t)__builtin_amdgcn_readlane((int)w, k);
    return;
  }
#endif
  (void)lane;
  for (int k = 0; k < 6; k++) hd[k] = ld_u32(R + 4 * k);
}
// the CIGAR operations [base + lane * OPL, + OPL) of a record; operations past the end read as 0P.  Wave form: one 16-byte
// load per lane (64 lanes: 1 KB of consecutive operations, sixteen 64-byte lines per instruction).
template <bool WAVE> SNF_HD void x_load_ops(const uint8_t* cig, int n_cig, int base, int lane, uint32_t* dst) {
#if XDEV
  if (WAVE) {
    const int k0 = base + lane * 4;
    if (k0 + 4 <= n_cig) {
      const uint4 q = *(const x_u128_any*)(cig + 4 * (int64_t)k0);
      dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) dst[j] = k0 + j < n_cig ? *(const x_u32_any*)(cig + 4 * (int64_t)(k0 + j)) : 6u;
    }
    return;
  }
#endif
  dst[0] = base + lane < n_cig ? ld_u32(cig + 4 * (int64_t)(base + lane)) : 6u;
}
SNF_HD void x_error(const ExView& v, int64_t rec, int code) {
  unsigned long long w = ((unsigned long long)rec << 8) | (unsigned)code;
#if XDEV
  atomicMin(v.err, w);
#else
  if (w < *v.err) *v.err = w;
#endif
}
SNF_HD uint8_t x_base(uint32_t nib) {   // "=ACMGRSVTWYHKDBN"
  const uint64_t t0 = 0x565352474D43413DULL;   // '=' 'A' 'C' 'M' 'G' 'R' 'S' 'V'
  const uint64_t t1 = 0x4E42444B48595754ULL;   // 'T' 'W' 'Y' 'H' 'K' 'D' 'B' 'N'
  return (uint8_t)(((nib & 8) ? t1 : t0) >> ((nib & 7) * 8));
}
#define X_REFC 0x18Du    // M D N = X consume the reference
#define X_QRYC 0x193u    // M I S = X consume the read (OPTAB add_read, leadprov.py:182-192)
#define X_EVENT 0x016u   // I D S (OPTAB event column)

// offset of the first NUL in blob[p, end), -1 if none
template <bool WAVE> SNF_HD int64_t x_find_nul(const uint8_t* blob, int64_t p, int64_t end, int lane) {
  const int W = WAVE ? 64 : 1;
  for (int64_t q = p; q < end; q += W) {
    const bool z = (q + lane < end) && blob[q + lane] == 0;
    const uint64_t m = x_ballot<WAVE>(z);
    if (m) return q + x_ctz(m);
  }
  return -1;
}

// copy read bases [a, a + n) (4-bit packed at seq) as ASCII to dst, all lanes
template <bool WAVE> SNF_HD void x_copy_bases(const uint8_t* seq, int32_t a, int32_t n, uint8_t* dst, int lane) {
  const int W = WAVE ? 64 : 1;
  for (int32_t j = lane; j < n; j += W) {
    const int32_t i = a + j;
    const uint32_t b = seq[i >> 1];
    dst[j] = x_base((i & 1) ? (b & 15) : (b >> 4));
  }
}

// ---- SA string parsing (sequential; lane 0) --------------------------------------------------------------------
// `s` points at the first byte of the SA string - in the LDS copy of the record's auxiliary region (wave form) or in the blob;
// positions are byte offsets from there
struct SaElem { int32_t f0[6], f1[6]; int nf; };   // field byte ranges [f0, f1) of one SA element

// field number e->nf is [fs, b).  (Constant indices only: the element stays in registers - indexed by nf it lived in scratch memory,
// and the parser, a chain of dependent accesses on one lane, waited a memory round trip for every field it wrote and read back:
// a record with an SA tag took ~250 us, which is what a whole pass over 24 000 records then took)
SNF_HD void sa_field(SaElem* e, int32_t fs, int32_t b) {
#pragma unroll
  for (int q = 0; q < 6; q++) if (e->nf == q) { e->f0[q] = fs; e->f1[q] = b; }
  e->nf++;
}
// next non-empty element of the ';' separated string starting at *p (NUL terminated); false at the end
SNF_HD bool sa_next(const uint8_t* s, int32_t* p, SaElem* e) {
  for (;;) {
    int32_t a = *p;
    if (s[a] == 0) return false;
    // one pass: the element's end and its field borders (eight bytes per read)
    e->nf = 0;
    int32_t fs = a, b = a; bool end = false;
    while (!end) {
      uint64_t w = x_ld8(s + b);
      for (int k = 0; k < 8; k++, w >>= 8) {
        const uint8_t c = (uint8_t)w;
        if (c == 0 || c == ';') { end = true; break; }
        if (c == ',') { sa_field(e, fs, b); fs = b + 1; }
        b++;
      }
    }
    *p = s[b] == ';' ? b + 1 : b;
    if (b == a) continue;
    sa_field(e, fs, b);
    return true;
  }
}
SNF_HD bool sa_int(const uint8_t* s, int32_t a, int32_t b, int64_t* out) {
  bool neg = false;
  if (a < b && (s[a] == '-' || s[a] == '+')) { neg = s[a] == '-'; a++; }
  if (a >= b || b - a > 18) return false;
  int64_t v = 0;
  for (int32_t k = a; k < b; k += 8) {
    uint64_t w = x_ld8(s + k);
    const int m = b - k < 8 ? b - k : 8;
    for (int j = 0; j < m; j++, w >>= 8) { const uint8_t c = (uint8_t)w; if (c < '0' || c > '9') return false; v = v * 10 + (c - '0'); }
  }
  *out = neg ? -v : v;
  return true;
}
// CIGAR_analyze (leadprov.py:144-178); false = the reference raises inside its try block
SNF_HD bool sa_cigar(const uint8_t* s, int32_t a, int32_t b, int64_t* clip0, int64_t* clip1, int64_t* refspan, int64_t* readspan) {
  int64_t num = 0, rd = 0, rf = 0, clip = 0, first = -1; bool have = false;
  for (int32_t k0 = a; k0 < b; k0 += 8) {
    uint64_t w = x_ld8(s + k0);
    const int m = b - k0 < 8 ? b - k0 : 8;
    for (int j = 0; j < m; j++, w >>= 8) {
      const uint8_t c = (uint8_t)w;
      if (c >= '0' && c <= '9') { num = num * 10 + (c - '0'); have = true; continue; }
      if (!have) return false;
      const bool q = c == 'M' || c == 'I' || c == 'X' || c == '=';
      const bool r = c == 'M' || c == 'D' || c == 'X' || c == '=' || c == 'N';
      if (q) rd += num;
      if (r) rf += num;
      if (!q && !r) {
        if (c != 'S' && c != 'H') return false;
        if (first < 0 && rd + rf > 0) first = clip;
        clip += num;
      }
      num = 0; have = false;
    }
  }
  if (first < 0) first = clip;
  *clip0 = first; *clip1 = clip - first; *refspan = rf; *readspan = rd;
  return true;
}
SNF_HD int32_t sa_contig(const ExView& v, const uint8_t* s, int32_t a, int32_t b) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (int32_t k0 = a; k0 < b; k0 += 8) {
    uint64_t w = x_ld8(s + k0);
    const int m = b - k0 < 8 ? b - k0 : 8;
    for (int j = 0; j < m; j++, w >>= 8) h = (h ^ (uint8_t)w) * 0x100000001b3ULL;
  }
  int lo = 0, hi = v.n_contigs;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (v.ctg_hash[mid] < h) lo = mid + 1; else hi = mid; }
  return (lo < v.n_contigs && v.ctg_hash[lo] == h) ? v.ctg_rank[lo] : -1;
}
SNF_HD bool fits_i32(int64_t x) { return x >= INT32_MIN + 1 && x <= INT32_MAX; }

// ---- one lead row ------------------------------------------------------------------------------------------------
struct LeadRow {
  int32_t ref_start, ref_end, qry_start, qry_end, svlen, read_len, ps, mate_contig, mate_pos, seq_len;
  int64_t seq_off; double nm; uint8_t svtype, strand, mapq, source, hap, is_sa, first, rev;
};
SNF_HD void put_lead(const ExView& v, int64_t i, const LeadRow& r, uint32_t qname, uint32_t read_id) {
  v.o_ref_start[i] = r.ref_start; v.o_ref_end[i] = r.ref_end; v.o_qry_start[i] = r.qry_start; v.o_qry_end[i] = r.qry_end;
  v.o_svlen[i] = r.svlen; v.o_read_len[i] = r.read_len; v.o_ps[i] = r.ps; v.o_mate_contig[i] = r.mate_contig;
  v.o_mate_pos[i] = r.mate_pos; v.o_seq_len[i] = r.seq_len; v.o_seq_off[i] = r.seq_off; v.o_nm[i] = r.nm;
  v.o_qname[i] = qname; v.o_read_id[i] = read_id;
  v.o_svtype[i] = r.svtype; v.o_strand[i] = r.strand; v.o_mapq[i] = r.mapq; v.o_source[i] = r.source; v.o_hap[i] = r.hap;
  v.o_is_sa[i] = r.is_sa; v.o_first[i] = r.first; v.o_rev[i] = r.rev;
}

// classify_splits (sv.py:649-782) over segs[ord[0..n)); returns the (possibly filtered) count
SNF_HD int classify(const ExView& v, Seg* segs, uint8_t* ord, int n, int32_t l_seq, int64_t rec) {
  const int32_t m = v.cfg.minsvlen_screen;
  for (int pass = 0; pass < 2; pass++) {
    // stable insertion sort by query start
    for (int i = 1; i < n; i++) {
      const uint8_t x = ord[i]; int j = i;
      while (j > 0 && segs[ord[j - 1]].qs > segs[x].qs) { ord[j] = ord[j - 1]; j--; }
      ord[j] = x;
    }
    for (int i = 0; i < n; i++) segs[ord[i]].hint_type = -1;
    int hints = 0;
    { Seg& f = segs[ord[0]];
      if ((double)f.qs >= (double)v.cfg.long_ins_length * 0.5) { f.hint_type = SNF_INS; f.hint_start = f.rs; f.hint_len = SNF_SVLEN_NONE; } }
    for (int i = 1; i < n; i++) {
      const Seg& a = segs[ord[i - 1]]; Seg& b = segs[ord[i]];
      if (a.contig != b.contig) continue;
      const bool fwd = !b.rev;
      const int64_t dq = (int64_t)b.qs - a.qe;
      if (a.rev == b.rev) {
        const int64_t dr = fwd ? (int64_t)b.rs - a.re : (int64_t)a.rs - b.re;
        const int32_t at = fwd ? b.rs : a.rs;
        if (dq >= m && dq - dr >= m) {
          if (dq <= v.cfg.dev_seq_cache_maxlen) {
            if (l_seq == 0) x_error(v, rec, XE_NOSEQ);
            const int32_t lo = a.qe < l_seq ? a.qe : l_seq, hi = b.qs < l_seq ? b.qs : l_seq;
            b.seq_a = lo; b.seq_n = hi > lo ? hi - lo : 0;
          } else b.seq_n = -1;
          if (!fits_i32(dq)) x_error(v, rec, XE_RANGE);
          b.hint_type = SNF_INS; b.hint_start = at; b.hint_len = (int32_t)dq; hints++;
        } else if (dr >= m && dr - dq >= m) {
          if (!fits_i32(dr)) x_error(v, rec, XE_RANGE);
          b.hint_type = SNF_DEL; b.hint_start = at; b.hint_len = (int32_t)-dr; hints++;
        } else if (dr <= 0) {
          if (-dr >= m) { if (!fits_i32(dr)) x_error(v, rec, XE_RANGE); b.hint_type = SNF_DUP; b.hint_start = at; b.hint_len = (int32_t)-dr; hints++; }
        }
      } else {
        const int64_t x = fwd ? a.rs : a.re, y = fwd ? b.rs : b.re;
        const int64_t lo = x < y ? x : y, hi = x < y ? y : x;
        if (hi - lo >= m) { if (!fits_i32(hi - lo)) x_error(v, rec, XE_RANGE); b.hint_type = SNF_INV; b.hint_start = (int32_t)lo; b.hint_len = (int32_t)(hi - lo); hints++; }
      }
    }
    if (hints || n <= 2) return n;
    // no hint: keep the splits on the contig / strand of the first one; exactly two left -> classify those again
    const Seg& l = segs[ord[0]];
    int k = 0;
    for (int i = 0; i < n; i++) if (segs[ord[i]].contig == l.contig && segs[ord[i]].rev == l.rev) ord[k++] = ord[i];
    n = k;
    if (n != 2) return n;
  }
  return n;
}

// ---- the tags of a record (first NM / HP / PS / SA), counting pass ------------------------------------------------
// `A` points at the record's auxiliary region (its LDS copy in the wave form, the blob otherwise), positions are byte offsets
// from there.  A tag is ONE 8-byte read: two name bytes, the type, and up to five bytes of value (the sub-type and count of a
// B array); only Z / H strings are looked at further (the NUL, 64 bytes per step by the whole wave).
struct TagSum { int64_t ps; int32_t sa_rel, sa_len, nm_tag; int hp; bool has_nm, has_ps; int bad; };
template <bool WAVE> SNF_HD void x_parse_tags(const uint8_t* A, int32_t aux_len, int lane, TagSum& t) {
  t.ps = 0; t.sa_rel = -1; t.sa_len = 0; t.nm_tag = 0; t.hp = 0; t.has_nm = false; t.has_ps = false;
  bool has_hp = false; int bad = 0;
  int64_t p = 0;
  while (p + 3 <= aux_len && !bad) {
    // (every lane walks the same tags: with the word in scalar registers the walk is scalar code - branches on SCC instead of
    // saved and restored execution masks, which is what the compiler makes of a `switch` on a value it must assume differs per lane)
    const uint64_t w = x_uni64<WAVE>(x_ld8(A + p));
    const uint8_t t0 = (uint8_t)w, t1 = (uint8_t)(w >> 8), ty = (uint8_t)(w >> 16);
    const uint32_t pay = (uint32_t)(w >> 24);   // value bytes 0..3 (byte 4 of a B array's count is w >> 56)
    p += 3;
    int64_t val = 0; bool isint = true; int64_t zs = -1, zn = 0;
    switch (ty) {
      case 'A': isint = false; p += 1; break;
      case 'c': val = (int8_t)pay; p += 1; break;
      case 'C': val = (uint8_t)pay; p += 1; break;
      case 's': val = (int16_t)pay; p += 2; break;
      case 'S': val = (uint16_t)pay; p += 2; break;
      case 'i': val = (int32_t)pay; p += 4; break;
      case 'I': val = pay; p += 4; break;
      case 'f': isint = false; p += 4; break;
      case 'Z': case 'H': {
        isint = false; zs = p;
        const int64_t z = x_find_nul<WAVE>(A, p, aux_len, lane);
        if (z < 0) bad = XE_AUX; else { zn = z - p; p = z + 1; }
        break; }
      case 'B': {
        isint = false;
        const uint8_t sub = (uint8_t)pay; const int64_t cnt = (int32_t)(uint32_t)(w >> 32);
        const int sz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
        if (!sz || cnt < 0) bad = XE_AUX; else p += 5 + cnt * sz;
        break; }
      default: bad = XE_AUX;
    }
    if (bad) break;
    if (t0 == 'N' && t1 == 'M' && !t.has_nm) { if (!isint) bad = XE_TAGTYPE; t.has_nm = true; t.nm_tag = (int32_t)val; if (!fits_i32(val)) bad = XE_RANGE; }
    else if (t0 == 'H' && t1 == 'P' && !has_hp) { if (!isint) bad = XE_TAGTYPE; has_hp = true; if (val < 0 || val > 2) bad = bad ? bad : XE_HP; t.hp = (int)val; }
    else if (t0 == 'P' && t1 == 'S' && !t.has_ps) { if (!isint) bad = XE_TAGTYPE; t.has_ps = true; t.ps = val; }
    else if (t0 == 'S' && t1 == 'A' && t.sa_rel < 0) { if (ty != 'Z') bad = XE_TAGTYPE; t.sa_rel = (int32_t)zs; t.sa_len = (int32_t)zn; }
  }
  if (!bad && p > aux_len) bad = XE_AUX;
  t.bad = bad;
}

// ---- one alignment record ----------------------------------------------------------------------------------------
#ifndef X_AHEAD
#define X_AHEAD 2      /* measured: 2 = 3 = 4 steps ahead (0.108 / 0.111 / 0.114 ms per counting pass); the walk is not what a pass waits for */
#endif
#define XAUXCAP 1024   // bytes of a record's auxiliary region (counting pass) / of its SA string (emit pass) the wave form keeps in LDS;
                       // a longer one (base-modification arrays ...) is parsed where it lies in the blob
template <bool WAVE, bool EMIT>
SNF_HD void extract_record(int64_t rec, const ExView& v, Seg* segs, uint8_t* ord, uint8_t* auxl, int32_t* sab) {
  const int lane = x_lane<WAVE>();
  const int W = WAVE ? 64 : 1;
  const snf_extract_config_t& cfg = v.cfg;
  XT_DECL
  const int64_t roff = v.rec_off[rec];
  const uint8_t* R = v.blob + roff;
  RecSum& S = v.sum[rec];
  RecSum sv{};   // emit pass: what the counting pass left, requested together with the fixed fields (wave-uniform: into scalar registers)
  if (EMIT) {
    sv = S;
    sv.accept = (uint8_t)x_uni<WAVE>(sv.accept); sv.sa_rel = (int32_t)x_uni<WAVE>((uint32_t)sv.sa_rel); sv.sa_len = (int32_t)x_uni<WAVE>((uint32_t)sv.sa_len);
    sv.walk_lo = (int32_t)x_uni<WAVE>((uint32_t)sv.walk_lo); sv.walk_hi = (int32_t)x_uni<WAVE>((uint32_t)sv.walk_hi);
    sv.walk_pr = x_uni<WAVE>(sv.walk_pr); sv.walk_pf = x_uni<WAVE>(sv.walk_pf);
    sv.nm_tag = (int32_t)x_uni<WAVE>((uint32_t)sv.nm_tag); sv.ref_end = (int32_t)x_uni<WAVE>((uint32_t)sv.ref_end);
    sv.hp = (uint8_t)x_uni<WAVE>(sv.hp); sv.has_nm = (uint8_t)x_uni<WAVE>(sv.has_nm); sv.has_ps = (uint8_t)x_uni<WAVE>(sv.has_ps);
  }
  else if (lane == 0) { S.accept = 0; S.n_leads = 0; S.seq_bytes = 0; S.has_nm = 0; S.has_ps = 0; S.nm = -1.0; S.sa_rel = -1; S.sa_len = 0; S.hp = 0; S.ps = 0; S.nm_tag = 0; S.ref_end = 0; }
  uint32_t hd[6];
  x_header<WAVE>(R, lane, hd);
  if (EMIT) { if (!sv.accept) return; }
  const int32_t block_size = (int32_t)hd[0];
  const int32_t ref_id = (int32_t)hd[1], pos = (int32_t)hd[2];
  const int l_name = (int)(hd[3] & 0xffu), mapq = (int)((hd[3] >> 8) & 0xffu);
  const int n_cig = (int)(hd[4] & 0xffffu), flag = (int)(hd[4] >> 16);
  const int32_t l_seq = (int32_t)hd[5];
  if (ref_id != v.region_ref_id || (flag & 0x4)) return;
  if (n_cig == 0) { if (!EMIT && lane == 0) x_error(v, rec, XE_NOCIGAR); return; }
  const uint8_t* cig = R + 36 + l_name;
  const uint8_t* seq = cig + 4 * (int64_t)n_cig;
  const int64_t aux0 = (int64_t)(seq - v.blob) + (l_seq + 1) / 2 + l_seq, aux1 = roff + 4 + block_size;
  const int32_t aux_len = (int32_t)(aux1 - aux0);
  // ---- everything the record needs next is requested here, in one round trip:
  // (a) the first and the last four CIGAR operations (lanes 0..3: operation k; lanes 4..7: operation n_cig - 1 - (lane - 4), while
  //     that is >= 1): the clip scans of getQueryStart / getQueryEnd and the break side of Lead.for_bnd
  uint32_t cw = 6u;   // 0P: ends a clip scan
#if XDEV
  if (WAVE) {
    const int k = lane < 4 ? lane : n_cig - 1 - (lane - 4);
    if (lane < 4 ? k < n_cig : (lane < 8 && k >= 1)) cw = *(const x_u32_any*)(cig + 4 * k);
  }
#endif
  // (b) counting pass: the auxiliary region; emit pass: the SA string - 16 bytes per lane into LDS
  const uint8_t* A = v.blob + aux0;   // where the tags are parsed
  const int32_t sa_rel_in = EMIT ? sv.sa_rel : -1, sa_len_in = EMIT ? sv.sa_len : 0;
  const uint8_t* SAs = EMIT ? v.blob + aux0 + sa_rel_in : nullptr;   // first byte of the SA string
#if XDEV
  if (WAVE) {
    x_wave_sync<WAVE>();     // the record before is through with the LDS copy
    const int64_t src = EMIT ? aux0 + sa_rel_in : aux0;
    const int32_t nb = EMIT ? (sa_rel_in >= 0 ? sa_len_in + 1 : 0) : aux_len;
    if (nb <= XAUXCAP) {
      for (int32_t o = lane * 16; o < nb; o += 64 * 16) *(uint4*)(auxl + o) = *(const x_u128_any*)(v.blob + src + o);
      if (EMIT) SAs = auxl; else A = auxl;
    }
  }
#endif
  // (c) the first steps of the CIGAR walk
  constexpr int OPL = WAVE ? 4 : 1;
  const int STEP = W * OPL;
  int base0 = 0, base1 = n_cig;
  if (EMIT) { base0 = sv.walk_lo; base1 = sv.walk_hi; }
  constexpr int XD = WAVE ? X_AHEAD : 1;      // steps requested ahead of the one that is worked on
  uint32_t cur[OPL], nx[XD][OPL];
#pragma unroll
  for (int j = 0; j < OPL; j++) { cur[j] = 6u; for (int d = 0; d < XD; d++) nx[d][j] = 6u; }
  if (base0 < base1) x_load_ops<WAVE>(cig, n_cig, base0, lane, cur);
#pragma unroll
  for (int d = 0; d + 1 < XD; d++) if (base0 + (d + 1) * STEP < base1) x_load_ops<WAVE>(cig, n_cig, base0 + (d + 1) * STEP, lane, nx[d]);
  // pysam getQueryStart / getQueryEnd
  int32_t q0 = 0, q1 = l_seq; bool clip_bad = false;
  uint32_t c_first, c_last;
  {
    int k = 0; bool more = true;
#if XDEV
    if (WAVE) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!more) continue;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cw, j); const int op = c & 15;
        if (op == 5) { if (q0 != 0 && q0 != l_seq) clip_bad = true; }
        else if (op == 4) q0 += (int32_t)(c >> 4);
        else more = false;
      }
      k = 4;
      c_first = (uint32_t)__builtin_amdgcn_readlane((int)cw, 0);
      c_last = n_cig >= 2 ? (uint32_t)__builtin_amdgcn_readlane((int)cw, 4) : c_first;
    } else
#endif
    { c_first = ld_u32(cig); c_last = ld_u32(cig + 4 * (int64_t)(n_cig - 1)); }
    for (; more && k < n_cig; k++) {
      const uint32_t c = ld_u32(cig + 4 * k); const int op = c & 15;
      if (op == 5) { if (q0 != 0 && q0 != l_seq) clip_bad = true; }
      else if (op == 4) q0 += (int32_t)(c >> 4);
      else break;
    }
  }
  if (l_seq == 0) {
    q1 = 0;   // no sequence: length from the CIGAR (M I = X, and S while nothing was counted yet); rare, lane-serial
    for (int k = 0; k < n_cig; k++) {
      const uint32_t c = ld_u32(cig + 4 * k); const int op = c & 15;
      if (op == 0 || op == 1 || op == 7 || op == 8 || (op == 4 && q1 == 0)) q1 += (int32_t)(c >> 4);
    }
  } else {
    int k = n_cig - 1; bool more = true;
#if XDEV
    if (WAVE) {
#pragma unroll
      for (int j = 0; j < 4; j++) {   // (a lane whose operation would be operation 0, or none, holds 0P)
        if (!more) continue;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cw, 4 + j); const int op = c & 15;
        if (op == 5) { if (q1 != l_seq) clip_bad = true; }
        else if (op == 4) q1 -= (int32_t)(c >> 4);
        else more = false;
      }
      k = n_cig - 5;
    }
#endif
    for (; more && k >= 1; k--) {
      const uint32_t c = ld_u32(cig + 4 * k); const int op = c & 15;
      if (op == 5) { if (q1 != l_seq) clip_bad = true; }
      else if (op == 4) q1 -= (int32_t)(c >> 4);
      else break;
    }
  }
  const int32_t qal = q1 - q0;
  // iter_region filter (leadprov.py:493-501).  pysam evaluates the clip properties first: a bad clip raises there.
  if (clip_bad) { if (!EMIT && lane == 0) x_error(v, rec, XE_CLIP); return; }
  if (mapq < cfg.mapq || (flag & 0x100) || qal < cfg.min_alignment_length) return;
  if (cfg.exclude_flags >= 0 && (flag & cfg.exclude_flags)) return;
  if (pos < v.region_start || pos >= v.region_end) return;
  // ---- tags (first NM / HP / PS / SA)
  int32_t sa_rel = -1, sa_len = 0; int64_t ps = 0; int32_t nm_tag = 0; int hp = 0; bool has_nm = false, has_ps = false;
  if (!EMIT) {
    x_wave_sync<WAVE>();     // the LDS copy is complete
    TagSum t;
    x_parse_tags<WAVE>(A, aux_len, lane, t);
    if (t.bad) { if (lane == 0) x_error(v, rec, t.bad); return; }
    sa_rel = t.sa_rel; sa_len = t.sa_len; ps = t.ps; nm_tag = t.nm_tag; hp = t.hp; has_nm = t.has_nm; has_ps = t.has_ps;
    if (sa_rel >= 0) SAs = A + sa_rel;
  } else {
    sa_rel = sa_rel_in; sa_len = sa_len_in; ps = sv.ps; nm_tag = sv.nm_tag; hp = sv.hp; has_nm = sv.has_nm; has_ps = sv.has_ps;
  }
  XT(1)
  const bool has_sa = sa_rel >= 0;
  const bool supp = (flag & 0x800) != 0, rev = (flag & 0x10) != 0;
  const bool use_clips = cfg.detect_large_ins && !supp && !has_sa;
  const double half_long = (double)cfg.long_ins_length / 2.0;
  int32_t ps_rank = v.ps_null_rank;
  int64_t lead_base = 0, seq_base = 0; uint32_t read_id = 0, qname = 0; double nm = -1.0;
  if (EMIT) {
    if (has_ps) {
      int lo = 0, hi = v.n_ps;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (v.ps_val[mid] < ps) lo = mid + 1; else hi = mid; }
      ps_rank = v.ps_rank[lo];
    }
    lead_base = v.lead_off[rec]; seq_base = v.seq_off[rec];
    read_id = v.read_id_offset + (uint32_t)v.read_idx[rec] + 1;
    qname = v.qname_rank[rec]; nm = sv.nm;
  }
  // ---- read_iterindels: 256 CIGAR operations per step (4 consecutive operations per lane), X_AHEAD steps requested ahead
  uint32_t pr = 0, pf = (uint32_t)pos;
  int32_t lead_k = 0; int64_t seqb = 0; uint32_t large = 0; int bad_op = 0;
  int32_t w_lo = 0, w_hi = 0; uint32_t w_pr = 0, w_pf = 0;   // counting pass: the span of steps that produced leads
  if (EMIT) { pr = sv.walk_pr; pf = sv.walk_pf; }
  for (int base = base0; base < base1; base += STEP) {
#pragma unroll
    for (int j = 0; j < OPL; j++) nx[XD - 1][j] = 6u;
    if (base + XD * STEP < base1) x_load_ops<WAVE>(cig, n_cig, base + XD * STEP, lane, nx[XD - 1]);
    uint32_t aq_l = 0, ar_l = 0;
#pragma unroll
    for (int j = 0; j < OPL; j++) {
      const uint32_t op = cur[j] & 15, len = cur[j] >> 4;
      if (op > 8) bad_op = 1;
      const uint32_t opm = op > 8 ? 0u : (1u << op);
      aq_l += (X_QRYC & opm) ? len : 0u; ar_l += (X_REFC & opm) ? len : 0u;
      if ((op == 1 || op == 2) && len > 10) large += len;
    }
    const uint32_t iq = x_incl_scan<WAVE>(aq_l, lane), ir = x_incl_scan<WAVE>(ar_l, lane);
    const uint32_t lane_pr = pr + iq - aq_l, lane_pf = pf + ir - ar_l;
    // signature-bearing operations of this lane
    uint32_t cnt_l = 0, sb_l = 0;
    { uint32_t p_r = lane_pr, p_f = lane_pf;
#pragma unroll
      for (int j = 0; j < OPL; j++) {
        const uint32_t op = cur[j] & 15, len = cur[j] >> 4;
        const uint32_t opm = op > 8 ? 0u : (1u << op);
        if ((X_EVENT & opm) && (int64_t)len >= cfg.minsvlen_screen) {
          const int32_t rs = (int32_t)(op == 2 ? p_f + len : p_f);
          const bool out = rs >= v.region_start && rs < v.region_end;
          if (op == 1 && (int64_t)len <= cfg.dev_seq_cache_maxlen) {
            if (l_seq == 0) bad_op = 2;
            const uint32_t ls = (uint32_t)l_seq, lo = p_r < ls ? p_r : ls, hi = p_r + len < ls ? p_r + len : ls;
            if (out) sb_l += hi - lo;
          }
          cnt_l += out ? 1u : 0u;
        }
        p_r += (X_QRYC & opm) ? len : 0u; p_f += (X_REFC & opm) ? len : 0u;
      } }
    if (x_ballot<WAVE>(cnt_l != 0)) {
      if (!EMIT) { if (w_hi == 0) { w_lo = base; w_pr = pr; w_pf = pf; } w_hi = base + STEP; }
      const uint32_t ic = x_incl_scan<WAVE>(cnt_l, lane), isb = x_incl_scan<WAVE>(sb_l, lane);
      if (EMIT) {
        int64_t my_k = lead_base + lead_k + (ic - cnt_l), my_so = seq_base + seqb + (isb - sb_l);
        int32_t cp_a[OPL], cp_n[OPL]; int64_t cp_dst[OPL];
        uint32_t p_r = lane_pr, p_f = lane_pf;
#pragma unroll
        for (int j = 0; j < OPL; j++) {
          const uint32_t op = cur[j] & 15, len = cur[j] >> 4;
          const uint32_t opm = op > 8 ? 0u : (1u << op);
          cp_n[j] = 0; cp_a[j] = 0; cp_dst[j] = 0;
          if ((X_EVENT & opm) && (int64_t)len >= cfg.minsvlen_screen) {
            LeadRow r{};
            r.qry_start = (int32_t)p_r; r.qry_end = (int32_t)(p_r + len); r.ref_start = r.ref_end = (int32_t)p_f;
            r.seq_len = SNF_SEQ_NONE; r.source = SNF_SRC_INLINE;
            if (op == 1) {
              r.svtype = SNF_INS; r.svlen = (int32_t)len;
              if ((int64_t)len <= cfg.dev_seq_cache_maxlen) {
                const uint32_t ls = (uint32_t)l_seq, lo = p_r < ls ? p_r : ls, hi = p_r + len < ls ? p_r + len : ls;
                r.seq_len = (int32_t)(hi - lo); cp_a[j] = (int32_t)lo;
              }
            } else if (op == 2) {
              r.svtype = SNF_DEL; r.svlen = -(int32_t)len; r.ref_start = (int32_t)(p_f + len); r.qry_end = (int32_t)p_r;
            } else if (use_clips && (double)len >= half_long) {
              r.svtype = SNF_INS; r.svlen = SNF_SVLEN_NONE;
            } else {
              r.svtype = p_f == (uint32_t)pos ? SNF_SINGLE_LEFT : SNF_SINGLE_RIGHT; r.svlen = 0;
            }
            if (r.ref_start >= v.region_start && r.ref_start < v.region_end) {
              r.read_len = qal; r.ps = ps_rank; r.nm = nm; r.strand = rev; r.mapq = (uint8_t)mapq; r.hap = (uint8_t)hp; r.is_sa = supp;
              r.seq_off = r.seq_len >= 0 ? my_so : 0;
              put_lead(v, my_k++, r, qname, read_id);
              if (r.seq_len > 0) { cp_n[j] = r.seq_len; cp_dst[j] = my_so; my_so += r.seq_len; }
            }
          }
          p_r += (X_QRYC & opm) ? len : 0u; p_f += (X_REFC & opm) ? len : 0u;
        }
        // inserted bases: every lane's pending slices are decoded by the whole wave
#pragma unroll
        for (int j = 0; j < OPL; j++) {
          uint64_t cm = x_ballot<WAVE>(cp_n[j] > 0);
          while (cm) {
            const int src = x_ctz(cm); cm &= cm - 1;
            const int32_t ca = (int32_t)x_bcast<WAVE>((uint32_t)cp_a[j], src), cn = (int32_t)x_bcast<WAVE>((uint32_t)cp_n[j], src);
            const int64_t dst = x_bcast64<WAVE>(cp_dst[j], src);
            x_copy_bases<WAVE>(seq, ca, cn, v.o_pool + dst, lane);
          }
        }
      }
      lead_k += (int32_t)x_bcast<WAVE>(ic, W - 1); seqb += x_bcast<WAVE>(isb, W - 1);
    }
    pr += x_bcast<WAVE>(iq, W - 1); pf += x_bcast<WAVE>(ir, W - 1);
#pragma unroll
    for (int j = 0; j < OPL; j++) {
      cur[j] = nx[0][j];
#pragma unroll
      for (int d = 0; d + 1 < XD; d++) nx[d][j] = nx[d + 1][j];
    }
  }
  if (!EMIT && x_ballot<WAVE>(bad_op != 0)) {
    const uint64_t b2 = x_ballot<WAVE>(bad_op == 2);
    if (lane == 0) x_error(v, rec, b2 ? XE_NOSEQ : XE_CIGAROP);
    return;
  }
  const int32_t ref_end = EMIT ? sv.ref_end : (pf == (uint32_t)pos ? pos + 1 : (int32_t)pf);
  if (!EMIT) {
    const uint32_t il = x_incl_scan<WAVE>(large, lane);
    const uint32_t large_sum = x_bcast<WAVE>(il, W - 1);
    if (cfg.advanced_tags && has_nm) nm = (double)((int64_t)nm_tag - (int64_t)large_sum) / (double)(qal + 1);
  } else if (lane == 0) {
    const int64_t ri = v.read_idx[rec];
    v.o_rstart[ri] = pos; v.o_rend[ri] = ref_end; v.o_rhp[ri] = (uint8_t)hp;
  }
  // ---- supplementary alignments: Lead.for_bnd, read_itersplits
  // Part A turns the SA string into the break-end lead of its first element and the segment table of classify_splits; part B
  // (lane 0: a handful of segments) classifies and emits.  Wave form of part A: the string is cut into its elements by the whole
  // wave (a byte per lane: ballots give every non-empty element its number, its first byte and its terminator), then EVERY
  // ELEMENT IS PARSED BY A LANE OF ITS OWN - the checks of the reference's loop are evaluated per element and the first element
  // that fails decides, as the loop's `break` does.  (Lane 0 walking the string byte by byte, three times over, took 35 us per
  // element - 330 us for a read with nine - and a pass over 24 000 records was as long as its slowest record.)
  int n_seg = 0, n_raw = 0, err = 0;
  if (EMIT && has_sa) x_wave_sync<WAVE>();   // the LDS copy of the SA string is complete
#ifdef SNF_XTRACE
  const unsigned long long xt_sa0 = wall_clock64();
#endif
  // for_bnd: side of the break from the first / last CIGAR operation
  const int64_t clip_l = ((c_first & 15) == 4 || (c_first & 15) == 5) ? (c_first >> 4) : 0, clip_r = ((c_last & 15) == 4 || (c_last & 15) == 5) ? (c_last >> 4) : 0;
  const bool is_first = !(clip_l > clip_r);
  const int32_t b_start = is_first ? ref_end : pos + 1;
  auto primary_seg = [&]() {
    Seg& s0 = segs[0];
    s0.contig = v.region_rank; s0.rs = pos; s0.re = ref_end; s0.qs = rev ? l_seq - q1 : q0; s0.qe = s0.qs + qal; s0.mapq = mapq;
    s0.rev = rev; s0.source = SNF_SRC_SPLIT_PRIM; s0.seq_n = -1; s0.job_dst = -1; s0.hint_type = -1;
  };
  auto bnd_lead = [&](int32_t mc, int64_t mate, bool mate_rev, int64_t s_nm) {
    if (EMIT) {
      LeadRow r{};
      r.ref_start = r.ref_end = b_start; r.qry_start = q0; r.qry_end = q1; r.svlen = 0; r.read_len = 0; r.ps = SNF_PS_NONE;
      r.mate_contig = mc; r.mate_pos = (int32_t)mate; r.seq_len = SNF_SEQ_NONE;
      r.nm = has_nm ? (double)s_nm : __builtin_nan("");
      r.svtype = SNF_BND; r.strand = rev; r.mapq = (uint8_t)mapq; r.source = SNF_SRC_BND_SA; r.hap = 0; r.is_sa = 0;
      r.first = is_first; r.rev = mate_rev;
      put_lead(v, lead_base + lead_k, r, qname, read_id);
    }
    lead_k++;
  };
  bool part_a_done = false;
#if XDEV
  if (WAVE && has_sa) {
    part_a_done = true;
    const uint8_t* B = SAs;
    // -- the elements: first byte and terminator of every non-empty one (the NUL at sa_len ends the last like a ';')
    int32_t ns = 0, ne = 0;
    for (int32_t c0 = 0; c0 <= sa_len; c0 += 64) {
      const int32_t p = c0 + lane;
      const bool in = p <= sa_len;
      const uint8_t ch = (in && p < sa_len) ? B[p] : (uint8_t)';';
      const uint8_t pv = (in && p >= 1) ? B[p - 1] : (uint8_t)';';
      const bool st = in && ch != ';' && pv == ';', en = in && ch == ';' && pv != ';';
      const uint64_t ms = __ballot(st), me = __ballot(en), below = (1ull << lane) - 1ull;
      if (st) { const int e = ns + x_popc(ms & below); if (e < XMAXSEG) sab[e] = p; }
      if (en) { const int e = ne + x_popc(me & below); if (e < XMAXSEG) sab[XMAXSEG + e] = p; }
      ns += x_popc(ms); ne += x_popc(me);
    }
    const int n_el = ns;
    x_wave_sync<WAVE>();
    // -- element `lane`: its fields, parsed
    const bool mine = lane < n_el;
    int nf = 0; bool strand_ok = false, srev = false, pos_ok = false, mq_ok = false, nm_ok = false, cig_ok = false;
    int64_t s_pos = 0, s_mq = 0, s_nm = 0, cl0 = 0, cl1 = 0, rfs = 0, rds = 0; int32_t mc = -1;
    if (mine) {
      const int32_t ea = sab[lane], eb = sab[XMAXSEG + lane];
      int32_t cm[5] = {eb, eb, eb, eb, eb}; int nc = 0;
      for (int32_t k0 = ea; k0 < eb; k0 += 8) {
        const uint64_t w = x_ld8(B + k0), x = w ^ 0x2c2c2c2c2c2c2c2cull;
        uint64_t z = ~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x | 0x7f7f7f7f7f7f7f7full);   // 0x80 in every ',' byte
        if (eb - k0 < 8) z &= (1ull << (8 * (eb - k0))) - 1ull;
        for (; z; z &= z - 1ull) {
          const int32_t at = k0 + (x_ctz(z) >> 3);
#pragma unroll
          for (int q = 0; q < 5; q++) if (nc == q) cm[q] = at;
          nc++;
        }
      }
      nf = nc + 1;
      if (nf == 6) {
        const uint8_t sc = B[cm[1] + 1];
        strand_ok = cm[2] - (cm[1] + 1) == 1 && (sc == '+' || sc == '-'); srev = sc == '-';
        mq_ok = sa_int(B, cm[3] + 1, cm[4], &s_mq);
        cig_ok = sa_cigar(B, cm[2] + 1, cm[3], &cl0, &cl1, &rfs, &rds);
        pos_ok = sa_int(B, cm[0] + 1, cm[1], &s_pos);
        mc = sa_contig(v, B, ea, cm[0]);
        if (lane == 0 && has_nm) nm_ok = sa_int(B, cm[4] + 1, eb, &s_nm);
      }
    }
    // -- Lead.for_bnd: the first element
    if (lane == 0 && n_el >= 1) {
      if (nf != 6) err = XE_SA_FIELDS;
      else if (!strand_ok) err = XE_SA_STRAND;
      else if (srev != rev) {
        if (!pos_ok) err = XE_SA_NUMBER;
        else if (cig_ok) {
          const bool mate_rev = cl1 > cl0;
          const int64_t z = s_pos - 1, mate = mate_rev ? z + rfs : (is_first ? z + 1 : z + 2);
          if (has_nm && !nm_ok) err = XE_SA_NUMBER;
          else if (mc < 0) err = XE_SA_CONTIG;
          else if (!fits_i32(mate)) err = XE_RANGE;
          else if (b_start >= v.region_start && b_start < v.region_end) bnd_lead(mc, mate, mate_rev, s_nm);
        }
      }
    }
    // -- read_itersplits: primary alignments only; every element a segment
    const bool stop = x_bcast<WAVE>((uint32_t)err, 0) != 0 || supp;
    if (!stop && !((double)n_el > (double)cfg.max_splits_base + cfg.max_splits_kb * ((double)l_seq / 1000.0))) {
      if (n_el + 1 > XMAXSEG) err = XE_SPLITS;
      else {
        int st = 0;   // this element in the order the reference's loop checks it: an error code, -1 = the read's splits are dropped
        int64_t z = 0, sq = 0;
        if (mine) {
          z = s_pos - 1; sq = srev ? cl1 : cl0;
          if (nf != 6) st = XE_SA_FIELDS;
          else if (!mq_ok) st = XE_SA_NUMBER;
          else if (!strand_ok) st = XE_SA_STRAND;
          else if (!cig_ok) st = -1;
          else if (!pos_ok) st = XE_SA_NUMBER;
          else if (mc < 0) st = XE_SA_CONTIG;
          else if (!fits_i32(z + rfs) || !fits_i32(sq + rds) || s_mq < 0 || s_mq > 255) st = XE_RANGE;
        }
        const uint64_t bad = __ballot(st != 0);
        if (bad) {   // the first element that fails ends the loop: an error, or no splits for this read
          const int st1 = (int)x_bcast<WAVE>((uint32_t)st, x_ctz(bad));
          if (st1 > 0) err = st1;
        } else {
          if (lane == 0) primary_seg();
          if (mine) {
            Seg& sg = segs[1 + lane];
            sg.contig = mc; sg.rs = (int32_t)z; sg.re = (int32_t)(z + rfs); sg.qs = (int32_t)sq; sg.qe = (int32_t)(sq + rds); sg.mapq = (int32_t)s_mq;
            sg.rev = srev; sg.source = SNF_SRC_SPLIT_SUP; sg.seq_n = -1; sg.job_dst = -1; sg.hint_type = -1;
          }
          n_seg = n_el + 1;
        }
      }
    }
    x_wave_sync<WAVE>();
  }
#endif
  if (!part_a_done && has_sa && lane == 0) {   // thread form: the reference's loops as they are
    const uint8_t* B = SAs;
    SaElem e;
    int32_t p = 0;
    if (sa_next(B, &p, &e)) {
      int64_t s_pos = 0, s_nm = 0, cl0, cl1, rfs, rds;
      if (e.nf != 6) err = XE_SA_FIELDS;
      else if (e.f1[2] - e.f0[2] != 1 || (B[e.f0[2]] != '+' && B[e.f0[2]] != '-')) err = XE_SA_STRAND;
      else if ((B[e.f0[2]] == '-') != rev) {
        const int32_t mc = sa_contig(v, B, e.f0[0], e.f1[0]);
        if (!sa_int(B, e.f0[1], e.f1[1], &s_pos)) err = XE_SA_NUMBER;
        else if (sa_cigar(B, e.f0[3], e.f1[3], &cl0, &cl1, &rfs, &rds)) {
          const bool mate_rev = cl1 > cl0;
          const int64_t z = s_pos - 1, mate = mate_rev ? z + rfs : (is_first ? z + 1 : z + 2);
          if (has_nm && !sa_int(B, e.f0[5], e.f1[5], &s_nm)) err = XE_SA_NUMBER;
          else if (mc < 0) err = XE_SA_CONTIG;
          else if (!fits_i32(mate)) err = XE_RANGE;
          else if (b_start >= v.region_start && b_start < v.region_end) bnd_lead(mc, mate, mate_rev, s_nm);
        }
      }
    }
    // read_itersplits: primary alignments only
    if (!err && !supp) {
      int n_el = 0; p = 0;
      while (sa_next(B, &p, &e)) n_el++;
      if (!((double)n_el > (double)cfg.max_splits_base + cfg.max_splits_kb * ((double)l_seq / 1000.0))) {
        if (n_el + 1 > XMAXSEG) err = XE_SPLITS;
        else {
          primary_seg();
          n_seg = 1; p = 0; bool drop = false;
          while (!err && !drop && sa_next(B, &p, &e)) {
            int64_t s_pos = 0, s_mq = 0, cl0, cl1, rfs, rds;
            if (e.nf != 6) { err = XE_SA_FIELDS; break; }
            if (!sa_int(B, e.f0[4], e.f1[4], &s_mq)) { err = XE_SA_NUMBER; break; }
            if (e.f1[2] - e.f0[2] != 1 || (B[e.f0[2]] != '+' && B[e.f0[2]] != '-')) { err = XE_SA_STRAND; break; }
            if (!sa_cigar(B, e.f0[3], e.f1[3], &cl0, &cl1, &rfs, &rds)) { drop = true; break; }
            if (!sa_int(B, e.f0[1], e.f1[1], &s_pos)) { err = XE_SA_NUMBER; break; }
            const int32_t mc = sa_contig(v, B, e.f0[0], e.f1[0]);
            if (mc < 0) { err = XE_SA_CONTIG; break; }
            const bool srev = B[e.f0[2]] == '-';
            const int64_t z = s_pos - 1, sq = srev ? cl1 : cl0;
            if (!fits_i32(z + rfs) || !fits_i32(sq + rds) || s_mq < 0 || s_mq > 255) { err = XE_RANGE; break; }
            Seg& s = segs[n_seg];
            s.contig = mc; s.rs = (int32_t)z; s.re = (int32_t)(z + rfs); s.qs = (int32_t)sq; s.qe = (int32_t)(sq + rds); s.mapq = (int32_t)s_mq;
            s.rev = srev; s.source = SNF_SRC_SPLIT_SUP; s.seq_n = -1; s.job_dst = -1; s.hint_type = -1;
            n_seg++;
          }
          if (err || drop) n_seg = 0;
        }
      }
    }
  }
  // part B (lane 0): classify_splits over the segment table, the leads of the hints
  if (has_sa && lane == 0) {
    if (!err && n_seg) {
      for (int i = 0; i < n_seg; i++) ord[i] = (uint8_t)i;
      n_raw = n_seg;
      n_seg = classify(v, segs, ord, n_seg, l_seq, rec);
      for (int i = 0; i < n_seg; i++) {
        Seg& s = segs[ord[i]];
        if (s.hint_type < 0) continue;
        const int32_t pm = segs[ord[i > 0 ? i - 1 : 0]].mapq;
        if (!cfg.dev_keep_lowqual_splits && (s.mapq < pm ? s.mapq : pm) < cfg.mapq) continue;
        if (s.contig != v.region_rank || s.hint_start < v.region_start || s.hint_start >= v.region_end) continue;
        const bool ins = s.hint_type == SNF_INS;
        const int32_t sl = (ins && s.seq_n >= 0) ? s.seq_n : -1;
        if (EMIT) {
          LeadRow r{};
          r.ref_start = s.hint_start;
          r.ref_end = (s.hint_len != SNF_SVLEN_NONE && !ins) ? s.hint_start + s.hint_len : s.hint_start;
          r.qry_start = s.qs; r.qry_end = s.qe; r.svlen = s.hint_len; r.read_len = 0; r.ps = ps_rank; r.mate_contig = 0;
          r.seq_len = sl; r.seq_off = sl >= 0 ? seq_base + seqb : 0; r.nm = nm;
          r.svtype = (uint8_t)s.hint_type; r.strand = s.rev; r.mapq = (uint8_t)s.mapq; r.source = s.source; r.hap = (uint8_t)hp; r.is_sa = supp;
          put_lead(v, lead_base + lead_k, r, qname, read_id);
          if (sl > 0) s.job_dst = seq_base + seqb;
        }
        lead_k++;
        if (sl > 0) seqb += sl;
      }
    }
    if (err) { x_error(v, rec, err); n_raw = 0; lead_k = -1; }
  }
  XT_ADD(2, xt_sa0)
  if (!EMIT) {
    if (lane == 0 && lead_k >= 0) {
      S.accept = 1; S.n_leads = lead_k; S.seq_bytes = seqb; S.has_nm = has_nm; S.has_ps = has_ps; S.nm = nm; S.sa_rel = sa_rel; S.sa_len = sa_len;
      S.hp = (uint8_t)hp; S.ps = ps; S.nm_tag = nm_tag; S.ref_end = ref_end;
      S.walk_lo = w_lo; S.walk_hi = w_hi; S.walk_pr = w_pr; S.walk_pf = w_pf;
    }
  } else if (has_sa && !supp) {
    // sequence of split-read insertions: lane 0 left the copy jobs in the segment table
    x_wave_sync<WAVE>();
    n_raw = (int)x_bcast<WAVE>((uint32_t)n_raw, 0);
    for (int i = 0; i < n_raw; i++) {
      const Seg& s = segs[i];
      if (s.job_dst >= 0) x_copy_bases<WAVE>(seq, s.seq_a, s.seq_n, v.o_pool + s.job_dst, lane);
    }
    x_wave_sync<WAVE>();
  }
  XT_FLUSH();
}

SNF_HD void x_count_body(int64_t i, const ExView& v) { Seg segs[XMAXSEG]; uint8_t ord[XMAXSEG]; extract_record<false, false>(i, v, segs, ord, nullptr, nullptr); }
SNF_HD void x_emit_body(int64_t i, const ExView& v) { Seg segs[XMAXSEG]; uint8_t ord[XMAXSEG]; extract_record<false, true>(i, v, segs, ord, nullptr, nullptr); }
SNF_HD void x_prep_body(int64_t i, const ExView& v) {
  const bool a = i < v.n_records && v.sum[i].accept;
  v.c_acc[i] = a ? 1 : 0; v.c_leads[i] = a ? v.sum[i].n_leads : 0; v.c_seq[i] = a ? v.sum[i].seq_bytes : 0;
  if (i < v.n_records) { const bool p = a && v.sum[i].has_ps; v.c_psf[i] = p ? 1 : 0; v.c_ps[i] = p ? v.sum[i].ps : 0; }
  const bool m = i < v.n_records && a && v.cfg.advanced_tags != 0 && v.sum[i].has_nm;
  if (i < v.n_records) v.c_nm[i] = m ? v.sum[i].nm : 0.0;
#if XDEV
  const unsigned long long mm = __ballot(m);      // (one atomic per wave: the number of summands)
  if (mm && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(mm)) atomicAdd((unsigned long long*)&v.tot[4], (unsigned long long)__popcll(mm));
#else
  if (m) v.tot[4]++;
#endif
}
SNF_HD void x_totals_body(int64_t i, const ExView& v) {
  const int64_t n = v.n_records;
  v.tot[1] = (unsigned long long)v.read_idx[n]; v.tot[2] = (unsigned long long)v.lead_off[n]; v.tot[3] = (unsigned long long)v.seq_off[n];
}

}  // namespace snf
using namespace snf;
SNF_KERNEL(x_count, ExView)
SNF_KERNEL(x_emit, ExView)
SNF_KERNEL(x_prep, ExView)
SNF_KERNEL(x_totals, ExView)

// one wave per record and ONE wave per workgroup: a record with split alignments keeps its wave several times longer
// than a plain one (lane-0 section), and in a four-wave workgroup the three finished waves' slots stayed taken until the
// slow one was done - with SA tags on 20 % of the records most workgroups had one.  Grid-stride so that any record count
// fits one launch.
// MINW: waves per SIMD the register allocation leaves room for (what it costs in registers is spilled - the lane-0 SA section
// takes most of it); SNF_EXTRACT_WAVES selects the instance
template <bool EMIT, int MINW>
__global__ void __launch_bounds__(64, MINW) x_wave(const ExView v, int64_t n) {
  __shared__ Seg segs[XMAXSEG];
  __shared__ uint8_t ord[XMAXSEG];
  __shared__ alignas(16) uint8_t auxl[XAUXCAP + 16];   // (reads of up to 15 bytes past the last copied byte stay inside)
  __shared__ int32_t sab[2 * XMAXSEG];                 // first byte / terminator of the SA string's elements
  // workgroups start in index order: the records with the longest CIGARs first (a read of 85 steps that starts among the last ones is
  // what a pass over a small table ended with: 40 % of its span with a few hundred waves in flight)
  for (int64_t w = (int64_t)blockIdx.x; w < n; w += (int64_t)gridDim.x)
    extract_record<true, EMIT>((int64_t)v.order[w], v, segs, ord, auxl, sab);
}
// average_regional_nm needs the reads' NM ratios summed in BAM order (leadprov.py:533-534, 577): sequential fp64 adds.
// x_prep has laid the summands out densely (a record that does not count as +0.0).  Adding a zero is the identity here - the sum starts
// at +0.0 and no summand is -0.0 (an integer over a positive integer), so it is never -0.0 itself - which is why zeros may be left out.
// One wave: 1024 values per step (coalesced loads, the next step's in flight meanwhile), the non-zero ones compacted in order into LDS,
// then folded - a chain of dependent adds (~19 cycles each on a lone wave: the floor of this kernel) whose operands are requested sixteen
// ahead.  (64 records per step straight from the 64-byte summaries through 128 v_readlanes: 0.40 ms for 24 000 records - longer than
// either pass; now 0.1 ms.)
#define X_NM_CHUNK 1024
__global__ void __launch_bounds__(64) x_nmsum(const ExView v, int64_t n) {
  __shared__ double buf[2][X_NM_CHUNK + 32];
  const int lane = threadIdx.x;
  constexpr int PER = X_NM_CHUNK / 64;
  double sum = 0.0;
  double x[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) { const int64_t i = lane + 64 * j; x[j] = i < n ? v.c_nm[i] : 0.0; }
  int cur = 0;
  for (int64_t base = 0; base < n; base += X_NM_CHUNK) {
    // the chunk's summands that are not zero, in order (adding a zero is the identity - see above -, and four records in ten do not count
    // at all): row j of the chunk is the 64 values x[j] of the lanes; a ballot and a prefix count give every value its place
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const unsigned long long mk = __ballot(x[j] != 0.0);
      if (x[j] != 0.0) buf[cur][cnt + __popcll(mk & ((1ull << lane) - 1ull))] = x[j];
      cnt += __popcll(mk);
    }
    if (lane < 32) buf[cur][cnt + lane] = 0.0;      // (the fold reads sixteen at a time, sixteen ahead)
#pragma unroll
    for (int j = 0; j < PER; j++) { const int64_t i = base + X_NM_CHUNK + lane + 64 * j; x[j] = i < n ? v.c_nm[i] : 0.0; }
    x_wave_sync<true>();
    // sixteen summands in registers, the next sixteen requested before the first is added (left to itself the compiler reads two
    // values, waits for them, adds them, reads the next two: an LDS round trip per pair - 34 cycles per summand, 0.34 ms for 24 000)
    const double* B = buf[cur];
    double a[16], nx[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = B[j];
    for (int k0 = 0; k0 < cnt; k0 += 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) nx[j] = B[k0 + 16 + j];
#if XDEV
      asm volatile("" ::: "memory");      // the reads stay in front of the adds
#endif
#pragma unroll
      for (int j = 0; j < 16; j++) sum += a[j];
#pragma unroll
      for (int j = 0; j < 16; j++) a[j] = nx[j];
    }
    cur ^= 1;
  }
  if (lane == 0) v.nm_out[0] = sum;
}

// ================================================================================================= host side ====
// One grow-only device allocation carved into arrays.  (A run made some forty hipMalloc / hipFree calls - 0.5 ms of a 1.4-ms run;
// repeated runs over tables of similar size now allocate nothing.)
struct XSlab {
  uint8_t* base = nullptr; size_t cap = 0, used = 0; bool measuring = false;
  void measure() { measuring = true; used = 0; }
  void reserve() {      // after a measuring pass: room for what it added up
    const size_t need = used + 256;
    if (need > cap) {
      if (base) (void)hipFree(base);
      base = nullptr; cap = 0;
      const size_t want = need + need / 4;
      void* p = nullptr;
      if (hipMalloc(&p, want) != hipSuccess) snf::fail("hipMalloc failed (" + std::to_string(want) + " bytes)");
      base = (uint8_t*)p; cap = want;
    }
    measuring = false; used = 0;
  }
  template <class T> T* take(size_t n, size_t pad_bytes = 0) {
    used = (used + 255) & ~(size_t)255;
    T* p = measuring ? nullptr : (T*)(base + used);
    used += (n ? n : 1) * sizeof(T) + pad_bytes;
    return p;
  }
  void release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; used = 0; }
};

struct snf_extract {
  snf_extract_config_t cfg;
  int device = 0;
  std::vector<void*> dev;        // device allocations of the current input / run
  XSlab run_a, run_b;            // what a run needs before / after the scans: two grow-only device blocks carved into arrays
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t ev_prep = nullptr;
  size_t scan_tmp = 0; int64_t scan_n = -1;
  ExView v{};
  int64_t n_records = 0, blob_len = 0;
  // host results
  std::vector<int32_t> h_i32[10]; std::vector<uint32_t> h_u32[2]; std::vector<int64_t> h_seq_off; std::vector<double> h_nm;
  std::vector<uint8_t> h_u8[8]; std::vector<uint8_t> h_pool; std::vector<int32_t> h_rs, h_re; std::vector<uint8_t> h_rhp;
  std::vector<int64_t> h_ps_value;
  snf_extract_result_t res{};
  bool have_input = false, have_result = false, pulled = false;
  int64_t n_leads = 0, n_seq = 0, n_reads = 0;
  hipStream_t side = nullptr;   // the serial NM sum runs beside the scans, the host round trip and the emit pass
};

namespace {
thread_local std::string g_xerr;

void x_free(void* p) {
  (void)hipFree(p);
}
template <class T> T* x_alloc(std::vector<void*>& pool, size_t n, size_t pad_bytes = 0) {
  void* p = nullptr;
  const size_t bytes = (n ? n : 1) * sizeof(T) + pad_bytes;
  if (hipMalloc(&p, bytes) != hipSuccess) snf::fail("hipMalloc failed (" + std::to_string(bytes) + " bytes)");
  pool.push_back(p);
  return (T*)p;
}
void x_h2d(void* d, const void* h, size_t bytes) {
  if (!bytes) return;
  SNF_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
}
void x_d2h(void* h, const void* d, size_t bytes) {
  if (!bytes) return;
  SNF_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
}
template <class T> T* x_up(std::vector<void*>& pool, const T* h, size_t n, size_t pad_bytes = 0) {
  T* d = x_alloc<T>(pool, n, pad_bytes);
  x_h2d(d, h, n * sizeof(T));
  return d;
}
void x_release(std::vector<void*>& pool) { for (void* p : pool) x_free(p); pool.clear(); }

// rank of str(value) among the distinct phase sets plus the literal "NULL" (Python str order)
bool ps_str_less(int64_t a, int64_t b) { return std::to_string(a) < std::to_string(b); }

#define X_WAVES_DEFAULT 4
template <bool EMIT> void x_launch_wave(int waves, unsigned grid, const ExView& v, int64_t n) {
  if (waves >= 8) hipLaunchKernelGGL((x_wave<EMIT, 8>), dim3(grid), dim3(64), 0, 0, v, n);
  else if (waves >= 6) hipLaunchKernelGGL((x_wave<EMIT, 6>), dim3(grid), dim3(64), 0, 0, v, n);
  else if (waves == 5) hipLaunchKernelGGL((x_wave<EMIT, 5>), dim3(grid), dim3(64), 0, 0, v, n);
  else hipLaunchKernelGGL((x_wave<EMIT, 4>), dim3(grid), dim3(64), 0, 0, v, n);
}

int do_upload(snf_extract* x, const snf_extract_input_t* in) {
  if (!in || in->n_records < 0 || (in->n_records && (!in->records || !in->rec_off || !in->qname_rank))) snf::fail("extract input: null pointer");
  if (in->n_contigs < 0 || (in->n_contigs && (!in->contig_hash || !in->contig_rank))) snf::fail("extract input: contig table missing");
  for (int64_t i = 0; i < in->n_records; i++) {
    const int64_t a = in->rec_off[i], b = in->rec_off[i + 1];
    if (a < 0 || b < a + 36 || b > in->records_len) snf::fail("extract input: record offsets malformed at record " + std::to_string(i));
    int32_t bs; memcpy(&bs, in->records + a, 4);
    if ((int64_t)bs + 4 != b - a) snf::fail("extract input: block_size of record " + std::to_string(i) + " disagrees with rec_off");
    const uint8_t* R = in->records + a;
    uint16_t n_cig; int32_t l_seq; memcpy(&n_cig, R + 16, 2); memcpy(&l_seq, R + 20, 4);
    if (l_seq < 0 || 36 + (int64_t)R[12] + 4 * (int64_t)n_cig + ((int64_t)l_seq + 1) / 2 + l_seq > b - a)
      snf::fail("extract input: record " + std::to_string(i) + " shorter than its fixed fields say");
  }
  for (int i = 1; i < in->n_contigs; i++)
    if (in->contig_hash[i - 1] >= in->contig_hash[i]) snf::fail("extract input: contig_hash must be strictly ascending");
  x_release(x->dev);
  x->have_result = false;
  ExView& v = x->v;
  v = ExView{};
  v.cfg = x->cfg;
  v.blob = x_up(x->dev, in->records, (size_t)in->records_len, 16);
  v.rec_off = x_up(x->dev, in->rec_off, (size_t)in->n_records + 1);
  v.qname_rank = x_up(x->dev, in->qname_rank, (size_t)in->n_records);
  v.ctg_hash = x_up(x->dev, in->contig_hash, (size_t)in->n_contigs);
  v.ctg_rank = x_up(x->dev, in->contig_rank, (size_t)in->n_contigs);
  {  // dispatch order of the wave form: counting sort of the records by CIGAR length, longest first; records of other contigs last
    const int64_t n = in->n_records;
    std::vector<uint32_t> cnt(65538, 0), ord((size_t)(n ? n : 1));
    auto key = [&](int64_t i) -> uint32_t {
      const uint8_t* R = in->records + in->rec_off[i];
      int32_t rid; uint16_t nc; memcpy(&rid, R + 4, 4); memcpy(&nc, R + 16, 2);
      return rid == in->region_ref_id ? 65535u - nc : 65536u;
    };
    for (int64_t i = 0; i < n; i++) cnt[key(i) + 1]++;
    for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
    for (int64_t i = 0; i < n; i++) ord[cnt[key(i)]++] = (uint32_t)i;
    v.order = x_up(x->dev, ord.data(), (size_t)(n ? n : 1));
  }
  v.n_contigs = in->n_contigs; v.n_records = in->n_records;
  v.region_ref_id = in->region_ref_id; v.region_rank = in->region_rank; v.region_start = in->region_start; v.region_end = in->region_end;
  v.read_id_offset = in->read_id_offset;
  x->n_records = in->n_records; x->blob_len = in->records_len;
  // algorithmic input bytes: everything of a record of this contig except its sequence and quality bytes
  int64_t ab = 0;
  for (int64_t i = 0; i < in->n_records; i++) {
    const uint8_t* R = in->records + in->rec_off[i];
    int32_t rid, l_seq; memcpy(&rid, R + 4, 4); memcpy(&l_seq, R + 20, 4);
    ab += (rid == in->region_ref_id) ? (in->rec_off[i + 1] - in->rec_off[i]) - ((int64_t)l_seq + 1) / 2 - l_seq : 36;
  }
  x->res.algo_bytes = ab;
  x->have_input = true;
  return 0;
}

// the result columns stay in HBM after a run (snf_batch_add_task_device takes them from there); the host copies are made by
// the first snf_extract_result
void pull_result(snf_extract* x) {
  if (x->pulled) return;
  ExView& v = x->v;
  const size_t L = (size_t)x->n_leads; const int64_t n_seq = x->n_seq, n_reads = x->n_reads;
  // results to the host
  int32_t* const di32[10] = {v.o_ref_start, v.o_ref_end, v.o_qry_start, v.o_qry_end, v.o_svlen, v.o_read_len, v.o_ps, v.o_mate_contig, v.o_mate_pos, v.o_seq_len};
  for (int k = 0; k < 10; k++) { x->h_i32[k].resize(L + 1); x_d2h(x->h_i32[k].data(), di32[k], L * 4); }
  x->h_u32[0].resize(L + 1); x_d2h(x->h_u32[0].data(), v.o_qname, L * 4);
  x->h_u32[1].resize(L + 1); x_d2h(x->h_u32[1].data(), v.o_read_id, L * 4);
  x->h_seq_off.resize(L + 1); x_d2h(x->h_seq_off.data(), v.o_seq_off, L * 8);
  x->h_nm.resize(L + 1); x_d2h(x->h_nm.data(), v.o_nm, L * 8);
  uint8_t* const du8[8] = {v.o_svtype, v.o_strand, v.o_mapq, v.o_source, v.o_hap, v.o_is_sa, v.o_first, v.o_rev};
  for (int k = 0; k < 8; k++) { x->h_u8[k].resize(L + 1); x_d2h(x->h_u8[k].data(), du8[k], L); }
  x->h_pool.resize((size_t)n_seq + 1); x_d2h(x->h_pool.data(), v.o_pool, (size_t)n_seq);
  x->h_rs.resize((size_t)n_reads + 1); x->h_re.resize((size_t)n_reads + 1); x->h_rhp.resize((size_t)n_reads + 1);
  x_d2h(x->h_rs.data(), v.o_rstart, (size_t)n_reads * 4); x_d2h(x->h_re.data(), v.o_rend, (size_t)n_reads * 4); x_d2h(x->h_rhp.data(), v.o_rhp, (size_t)n_reads);
  SNF_HIP(hipDeviceSynchronize());
  snf_task_input_t& t = x->res.task;
  t.ref_start = x->h_i32[0].data(); t.ref_end = x->h_i32[1].data(); t.qry_start = x->h_i32[2].data(); t.qry_end = x->h_i32[3].data();
  t.svlen = x->h_i32[4].data(); t.read_len = x->h_i32[5].data(); t.ps_rank = x->h_i32[6].data(); t.mate_contig = x->h_i32[7].data();
  t.mate_ref_start = x->h_i32[8].data(); t.seq_len = x->h_i32[9].data(); t.qname_id = x->h_u32[0].data(); t.read_id = x->h_u32[1].data();
  t.seq_off = x->h_seq_off.data(); t.nm = x->h_nm.data();
  t.svtype = x->h_u8[0].data(); t.strand = x->h_u8[1].data(); t.mapq = x->h_u8[2].data(); t.source = x->h_u8[3].data();
  t.hap = x->h_u8[4].data(); t.is_sa = x->h_u8[5].data(); t.bnd_is_first = x->h_u8[6].data(); t.bnd_is_reverse = x->h_u8[7].data();
  t.seq_pool_len = n_seq; t.seq_pool = x->h_pool.data();
  t.n_reads = n_reads; t.read_start = x->h_rs.data(); t.read_end = x->h_re.data(); t.read_hp = x->h_rhp.data();
  x->pulled = true;
}

int do_run(snf_extract* x) {
  if (!x->have_input) snf::fail("snf_extract_run before snf_extract_upload");
  x->have_result = false;
  ExView& v = x->v;
  const int64_t n = x->n_records;
  const size_t N1 = (size_t)n + 1;
  if (x->scan_n != n) {      // temporary storage of the three scans (one after the other on the stream: shared)
    size_t need = 0;
    SNF_HIP(rocprim::exclusive_scan(nullptr, need, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, N1, rocprim::plus<int64_t>(), 0));
    x->scan_tmp = need; x->scan_n = n;
  }
  int64_t *d_read_idx = nullptr, *d_lead_off = nullptr, *d_seq_off = nullptr; void* d_tmp = nullptr;
  auto carve_a = [&](XSlab& s) {
    v.sum = s.take<RecSum>((size_t)n);
    v.tot = s.take<unsigned long long>(8); v.err = v.tot;
    v.c_acc = s.take<int64_t>(N1); v.c_leads = s.take<int64_t>(N1); v.c_seq = s.take<int64_t>(N1);
    v.c_ps = s.take<int64_t>((size_t)n); v.c_psf = s.take<uint8_t>((size_t)n); v.c_nm = s.take<double>((size_t)n);
    v.nm_out = s.take<double>(2);
    d_read_idx = s.take<int64_t>(N1); d_lead_off = s.take<int64_t>(N1); d_seq_off = s.take<int64_t>(N1);
    d_tmp = s.take<uint8_t>(x->scan_tmp);
#ifdef SNF_XTRACE
    v.xtrace = s.take<uint32_t>((size_t)n * 4);
#endif
  };
  x->run_a.measure(); carve_a(x->run_a); x->run_a.reserve(); carve_a(x->run_a);
  v.read_idx = d_read_idx; v.lead_off = d_lead_off; v.seq_off = d_seq_off;
  const unsigned long long none = ~0ull;
  { const unsigned long long init[8] = {none, 0, 0, 0, 0, 0, 0, 0}; x_h2d(v.tot, init, 64); }      // [0] first error, [4] NM summands (x_prep)
#ifdef SNF_XTRACE
  SNF_HIP(hipMemset(v.xtrace, 0, (size_t)(n ? n : 1) * 16));
  v.xtrace_emit = getenv("SNF_XTRACE_PASS") && !strcmp(getenv("SNF_XTRACE_PASS"), "emit");
#endif
  float ms_count = 0, ms_emit = 0;
  const bool thread_form = getenv("SNF_EXTRACT_THREAD") != nullptr;
  const int waves = getenv("SNF_EXTRACT_WAVES") ? atoi(getenv("SNF_EXTRACT_WAVES")) : X_WAVES_DEFAULT;
  for (hipEvent_t& e : x->ev) if (!e) SNF_HIP(hipEventCreate(&e));
  hipEvent_t e0 = x->ev[0], e1 = x->ev[1], e2 = x->ev[2], e3 = x->ev[3];
  const int64_t grid_cap = getenv("SNF_EXTRACT_GRID") ? std::max(1, atoi(getenv("SNF_EXTRACT_GRID"))) : (1 << 22);   // (a capped grid strides: measured, slower - the dispatcher balances unequal records better)
  const unsigned grid_w = (unsigned)std::min<int64_t>(n > 0 ? n : 1, grid_cap);
  // (every record's summary is initialised by the counting pass itself, whatever becomes of the record)
  SNF_HIP(hipEventRecord(e0, 0));
  if (n) {
    if (thread_form) hipLaunchKernelGGL(x_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v, n);
    else x_launch_wave<false>(waves, grid_w, v, n);
  }
  SNF_HIP(hipEventRecord(e1, 0));
  hipLaunchKernelGGL(x_prep, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, 0, v, n + 1);
  if (!x->side) SNF_HIP(hipStreamCreateWithFlags(&x->side, hipStreamNonBlocking));
  if (!x->ev_prep) SNF_HIP(hipEventCreateWithFlags(&x->ev_prep, hipEventDisableTiming));
  SNF_HIP(hipEventRecord(x->ev_prep, 0));
  SNF_HIP(hipStreamWaitEvent(x->side, x->ev_prep, 0));      // the serial sum runs beside the scans, the host round trip and the emit pass
  hipLaunchKernelGGL(x_nmsum, dim3(1), dim3(64), 0, x->side, v, n);
  {
    size_t need = x->scan_tmp;
    SNF_HIP(rocprim::exclusive_scan(d_tmp, need, (const int64_t*)v.c_acc, d_read_idx, (int64_t)0, N1, rocprim::plus<int64_t>(), 0));
    SNF_HIP(rocprim::exclusive_scan(d_tmp, need, (const int64_t*)v.c_leads, d_lead_off, (int64_t)0, N1, rocprim::plus<int64_t>(), 0));
    SNF_HIP(rocprim::exclusive_scan(d_tmp, need, (const int64_t*)v.c_seq, d_seq_off, (int64_t)0, N1, rocprim::plus<int64_t>(), 0));
  }
  hipLaunchKernelGGL(x_totals, dim3(1), dim3(64), 0, 0, v, (int64_t)1);
  unsigned long long tot[8] = {none, 0, 0, 0, 0, 0, 0, 0}; double nmv[2] = {0, 0};
  x_d2h(tot, v.tot, 64);
  const unsigned long long err = tot[0]; const int64_t n_reads = (int64_t)tot[1], n_leads = (int64_t)tot[2], n_seq = (int64_t)tot[3];
  if (err != none) {
    const int code = (int)(err & 0xff);
    snf::fail("alignment record " + std::to_string((long long)(err >> 8)) + ": " + XE_TEXT[code < 14 ? code : 4]);
  }
  // phase sets: distinct PS values of the accepted reads -> rank of str(value) in Python str order, "NULL" last
  std::vector<int64_t> h_ps((size_t)n); std::vector<uint8_t> h_psf((size_t)n);
  x_d2h(h_ps.data(), v.c_ps, (size_t)n * 8); x_d2h(h_psf.data(), v.c_psf, (size_t)n);
  std::vector<int64_t> vals;
  for (int64_t i = 0; i < n; i++) if (h_psf[(size_t)i]) vals.push_back(h_ps[(size_t)i]);
  std::sort(vals.begin(), vals.end());
  vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
  std::vector<int64_t> by_str(vals);
  std::sort(by_str.begin(), by_str.end(), ps_str_less);
  std::vector<int32_t> rank_of(vals.size());
  for (size_t r = 0; r < by_str.size(); r++) rank_of[(size_t)(std::lower_bound(vals.begin(), vals.end(), by_str[r]) - vals.begin())] = (int32_t)r;
  v.n_ps = (int32_t)vals.size(); v.ps_null_rank = (int32_t)vals.size();   // digits and '-' sort before 'N'
  x->h_ps_value = by_str; x->h_ps_value.push_back(0);
  // outputs
  const size_t L = (size_t)n_leads;
  int64_t* d_ps_val = nullptr; int32_t* d_ps_rank = nullptr;
  auto carve_b = [&](XSlab& s) {
    d_ps_val = s.take<int64_t>(vals.size()); d_ps_rank = s.take<int32_t>(rank_of.size());
    v.o_ref_start = s.take<int32_t>(L); v.o_ref_end = s.take<int32_t>(L); v.o_qry_start = s.take<int32_t>(L); v.o_qry_end = s.take<int32_t>(L);
    v.o_svlen = s.take<int32_t>(L); v.o_read_len = s.take<int32_t>(L); v.o_ps = s.take<int32_t>(L); v.o_mate_contig = s.take<int32_t>(L);
    v.o_mate_pos = s.take<int32_t>(L); v.o_seq_len = s.take<int32_t>(L); v.o_qname = s.take<uint32_t>(L); v.o_read_id = s.take<uint32_t>(L);
    v.o_seq_off = s.take<int64_t>(L); v.o_nm = s.take<double>(L);
    v.o_svtype = s.take<uint8_t>(L); v.o_strand = s.take<uint8_t>(L); v.o_mapq = s.take<uint8_t>(L); v.o_source = s.take<uint8_t>(L);
    v.o_hap = s.take<uint8_t>(L); v.o_is_sa = s.take<uint8_t>(L); v.o_first = s.take<uint8_t>(L); v.o_rev = s.take<uint8_t>(L);
    v.o_pool = s.take<uint8_t>((size_t)n_seq);
    v.o_rstart = s.take<int32_t>((size_t)n_reads); v.o_rend = s.take<int32_t>((size_t)n_reads); v.o_rhp = s.take<uint8_t>((size_t)n_reads);
  };
  x->run_b.measure(); carve_b(x->run_b); x->run_b.reserve(); carve_b(x->run_b);
  x_h2d(d_ps_val, vals.data(), vals.size() * 8); x_h2d(d_ps_rank, rank_of.data(), rank_of.size() * 4);
  v.ps_val = d_ps_val; v.ps_rank = d_ps_rank;
  SNF_HIP(hipEventRecord(e2, 0));
  if (n) {
    if (thread_form) hipLaunchKernelGGL(x_emit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v, n);
    else x_launch_wave<true>(waves, grid_w, v, n);
  }
  SNF_HIP(hipEventRecord(e3, 0));
  SNF_HIP(hipStreamSynchronize(x->side));
  SNF_HIP(hipDeviceSynchronize());
  SNF_HIP(hipEventElapsedTime(&ms_count, e0, e1)); SNF_HIP(hipEventElapsedTime(&ms_emit, e2, e3));
  x_d2h(nmv, v.nm_out, 8); nmv[1] = (double)tot[4];
#ifdef SNF_XTRACE
  if (const char* path = getenv("SNF_XTRACE_OUT")) {
    std::vector<uint32_t> t((size_t)n * 4);
    x_d2h(t.data(), v.xtrace, (size_t)n * 16);
    if (FILE* f = fopen(path, "wb")) { fwrite(t.data(), 16, (size_t)n, f); fclose(f); }
  }
#endif
  x->n_leads = n_leads; x->n_seq = n_seq; x->n_reads = n_reads; x->pulled = false;
  snf_extract_result_t& r = x->res;
  const int64_t ab = r.algo_bytes;
  r = snf_extract_result_t{};
  snf_task_input_t& t = r.task;
  t.n_leads = n_leads; t.seq_pool_len = n_seq; t.n_reads = n_reads;   // (array pointers: filled by the first snf_extract_result)
  t.n_tr = -1; t.ps_null_rank = v.ps_null_rank;
  const double cnt = nmv[1] > 1.0 ? nmv[1] : 1.0;
  t.qc_nm_threshold = nmv[0] / cnt;   // average_regional_nm = nm_sum / float(max(1, nm_count))
  r.n_ps = (int64_t)x->h_ps_value.size(); r.ps_value = x->h_ps_value.data();
  r.read_id = x->v.read_id_offset + (uint32_t)n_reads; r.read_count = n_reads;
  r.ms_count = ms_count; r.ms_emit = ms_emit; r.algo_bytes = ab + n_seq + n_seq / 2;
  x->have_result = true;
  return 0;
}
}  // namespace

#define X_TRY(stmt)                                                    \
  try { stmt; }                                                        \
  catch (const snf::Error& e) { g_xerr = e.msg; return 1; }            \
  catch (const std::exception& e) { g_xerr = e.what(); return 1; }     \
  return 0;

extern "C" {
const char* snf_extract_last_error(void) { return g_xerr.c_str(); }

int snf_extract_create(const snf_extract_config_t* cfg, int device, snf_extract_t** out) {
  if (!cfg || !out) { g_xerr = "snf_extract_create: null argument"; return 1; }
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { g_xerr = "no HIP device: the extraction kernels need a gfx950 GPU (there is no CPU fallback)"; return 1; }
  if (device < 0 || device >= nd) { g_xerr = "device index out of range"; return 1; }
  if (hipSetDevice(device) != hipSuccess) { g_xerr = "hipSetDevice failed"; return 1; }
  snf_extract* x = new snf_extract();
  x->cfg = *cfg; x->device = device;
  *out = x;
  return 0;
}
int snf_extract_upload(snf_extract_t* x, const snf_extract_input_t* in) {
  if (!x) { g_xerr = "null handle"; return 1; }
  if (hipSetDevice(x->device) != hipSuccess) { g_xerr = "hipSetDevice failed"; return 1; }
  X_TRY(do_upload(x, in))
}
int snf_extract_run(snf_extract_t* x) {
  if (!x) { g_xerr = "null handle"; return 1; }
  if (hipSetDevice(x->device) != hipSuccess) { g_xerr = "hipSetDevice failed"; return 1; }
  X_TRY(do_run(x))
}
int snf_extract_result(snf_extract_t* x, snf_extract_result_t* out) {
  if (!x || !out) { g_xerr = "null argument"; return 1; }
  if (!x->have_result) { g_xerr = "snf_extract_result before a successful snf_extract_run"; return 1; }
  if (hipSetDevice(x->device) != hipSuccess) { g_xerr = "hipSetDevice failed"; return 1; }
  try { pull_result(x); }
  catch (const snf::Error& e) { g_xerr = e.msg; return 1; }
  *out = x->res;
  return 0;
}
// the scalar part of the result (counts, ps table, NM threshold, timings) without copying the columns to the host: the array
// pointers of out->task are null unless an earlier snf_extract_result made the host copies
int snf_extract_result_meta(snf_extract_t* x, snf_extract_result_t* out) {
  if (!x || !out) { g_xerr = "null argument"; return 1; }
  if (!x->have_result) { g_xerr = "snf_extract_result_meta before a successful snf_extract_run"; return 1; }
  *out = x->res;
  return 0;
}
// the same result as DEVICE pointers (HBM of the handle's device): what snf_batch_add_task_device reads.  Valid until the
// next snf_extract_upload / snf_extract_run / snf_extract_destroy on the handle.
int snf_extract_device_view(snf_extract_t* x, snf_task_input_t* out, int* device) {
  if (!x || !out) { g_xerr = "null argument"; return 1; }
  if (!x->have_result) { g_xerr = "snf_extract_device_view before a successful snf_extract_run"; return 1; }
  const ExView& v = x->v;
  snf_task_input_t t = x->res.task;    // scalars (ps_null_rank, qc_nm_threshold, counts)
  t.n_leads = x->n_leads; t.seq_pool_len = x->n_seq; t.n_reads = x->n_reads;
  t.ref_start = v.o_ref_start; t.ref_end = v.o_ref_end; t.qry_start = v.o_qry_start; t.qry_end = v.o_qry_end; t.svlen = v.o_svlen;
  t.read_len = v.o_read_len; t.qname_id = v.o_qname; t.read_id = v.o_read_id; t.ps_rank = v.o_ps; t.mate_contig = v.o_mate_contig;
  t.mate_ref_start = v.o_mate_pos; t.seq_len = v.o_seq_len; t.seq_off = v.o_seq_off; t.nm = v.o_nm;
  t.svtype = v.o_svtype; t.strand = v.o_strand; t.mapq = v.o_mapq; t.source = v.o_source; t.hap = v.o_hap; t.is_sa = v.o_is_sa;
  t.bnd_is_first = v.o_first; t.bnd_is_reverse = v.o_rev;
  t.seq_pool = v.o_pool; t.read_start = v.o_rstart; t.read_end = v.o_rend; t.read_hp = v.o_rhp;
  *out = t;
  if (device) *device = x->device;
  return 0;
}
void snf_extract_destroy(snf_extract_t* x) {
  if (!x) return;
  x_release(x->dev); x->run_a.release(); x->run_b.release();
  for (hipEvent_t e : x->ev) if (e) (void)hipEventDestroy(e);
  if (x->ev_prep) (void)hipEventDestroy(x->ev_prep);
  if (x->side) (void)hipStreamDestroy(x->side);
  delete x;
}
}
