// snf_combine.hip - multi-sample combine: cluster.resolve_block_groups (cluster.py:356-390) with
// SVGroup.align_call (sv.py:280-289) and the running means of SVGroup.add_candidate (sv.py:297-318).
//
// One flush window = one problem: the greedy nearest-group assignment is sequential (every accepted candidate
// moves its group's means), windows are <= a few dozen candidates, and windows of different contigs / SV types are
// independent - so the batch dimension is the parallel one.  The edit distance is evaluated on demand, only for
// (group, candidate) pairs that pass the distance gates, with the bit-parallel Myers blocks of snf_myers.h: on the
// GPU one wave per window (lane 0 decides, all lanes align); the thread-per-window body is the emulation / reference
// form (SNF_COMBINE_THREAD=1 selects it on the GPU).
#include "snf_myers.h"
#include "snf_ctx.h"
#include "snf_group_call.h"
#include "../../include/sniffles_amd.h"

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

namespace snf {

struct CombineView {
  snf_config_t cfg;
  int64_t n_problems;
  const int32_t* svtype; const int32_t* n_cands; const int32_t* n_groups; const int32_t* n_words;  // per problem
  const int64_t* c_off;   // candidate arrays offset per problem (n_problems + 1)
  const int64_t* g_off;   // initial-group arrays offset per problem
  const int64_t* s_off;   // group state arrays offset per problem (capacity n_groups + n_cands each)
  const int64_t* w_off;   // included-sample bitset words offset per problem
  const int64_t* k_off;   // Myers carry scratch offset per problem
  const int32_t *pos, *svlen, *support, *sample_id, *mate_contig, *mate_pos;
  const int64_t* alt_off; const uint8_t* alt_pool;          // global offsets (n_total_cands + 1)
  const double *g_pos_mean, *g_len_mean, *g_mate_mean; const int32_t *g_size, *g_mate_contig;
  const int64_t* g_alt_off; const uint8_t* g_alt_pool;
  const int64_t* g_samples_off; const int32_t* g_samples;
  // scratch / state
  double *st_pos, *st_len, *st_mate; int32_t *st_size, *st_mctg; int64_t *st_alt_lo, *st_alt_hi; uint8_t* st_alt_src;
  uint64_t* st_bits; int32_t* order; int8_t* carry; int32_t* st_alist;
  uint64_t* ed_scratch; const int64_t* e_off;   // serial Myers block states per problem (ed_serial_scratch_words of its longest string)
  // flush windows: wn_off[p] = first window of problem p; win_cand has (windows + 1) entries per problem, stored at
  // index (wn_off[p] + p + w); win_bin / win_thr per window (thr < 0: nothing is flushed)
  const int64_t* wn_off; const int32_t* win_cand; const int32_t* win_bin; const double* win_thr;
  int32_t* out_group;
  unsigned long long* stats;   // [3][64 stripes][16]: alignments, bytes (len a + len b), full-matrix DP cells (len a * len b)
};

struct LessSupportDesc {
  const int32_t* support;
  SNF_HD bool operator()(int32_t a, int32_t b) const { return support[a] != support[b] ? support[a] > support[b] : a < b; }
};

// One problem (a flush window, or a whole chain of them) - the thread form (WAVE == false: emulation build and
// SNF_COMBINE_THREAD=1) and the gfx950 wave form share this body.  WAVE: one wave per problem, lane 0 ("lead") runs the
// sequential greedy assignment and, whenever a (group, candidate) pair passes the distance gates, all 64 lanes evaluate
// its edit distance together (ed_wave_pair: lane = 64-row block, anti-diagonal schedule) - the alignment is > 99 % of
// the work of a window with kilobase insertions, and one thread doing it serially takes tens of milliseconds per pair.
template <bool WAVE>
SNF_HD void combine_run(int64_t p, const CombineView& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = WAVE ? (int)(threadIdx.x & 63) : 0;
#else
  const int lane = 0;
#endif
  const bool lead = lane == 0;
  const snf_config_t& cfg = v.cfg;
  const int64_t c0 = v.c_off[p], g0 = v.g_off[p], s0 = v.s_off[p];
  const int ng0 = v.n_groups[p], nw = v.n_words[p], svtype = v.svtype[p];
  const int32_t *pos = v.pos + c0, *svlen = v.svlen + c0, *support = v.support + c0, *sample = v.sample_id + c0;
  const int32_t *mctg = v.mate_contig + c0, *mpos = v.mate_pos + c0;
  double *gpos = v.st_pos + s0, *glen = v.st_len + s0, *gmate = v.st_mate + s0;
  int32_t *gsize = v.st_size + s0, *gmc = v.st_mctg + s0, *alist = v.st_alist + s0;
  int64_t *galo = v.st_alt_lo + s0, *gahi = v.st_alt_hi + s0; uint8_t* gsrc = v.st_alt_src + s0;
  uint64_t* bits = v.st_bits + v.w_off[p];
  int32_t* order = v.order + c0;
  int8_t* carry = v.carry + v.k_off[p];
  uint64_t* ed_scr = v.ed_scratch + v.e_off[p];
  (void)carry; (void)ed_scr;
  int32_t* out = v.out_group + c0;
  int ng = ng0;   // group slots used so far (slot index == group number over the whole chain)
  int na = ng0;   // active groups: alist[0..na) in list order (kept groups first, then the window's new groups)
  if (lead) {
    for (int g = 0; g < ng0; g++) {
      gpos[g] = v.g_pos_mean[g0 + g]; glen[g] = v.g_len_mean[g0 + g]; gmate[g] = v.g_mate_mean[g0 + g];
      gsize[g] = v.g_size[g0 + g]; gmc[g] = v.g_mate_contig[g0 + g];
      galo[g] = v.g_alt_off[g0 + g]; gahi[g] = v.g_alt_off[g0 + g + 1]; gsrc[g] = 1;
      for (int w = 0; w < nw; w++) bits[(int64_t)g * nw + w] = 0;
      for (int64_t k = v.g_samples_off[g0 + g]; k < v.g_samples_off[g0 + g + 1]; k++) {
        int32_t sid = v.g_samples[k];
        bits[(int64_t)g * nw + (sid >> 6)] |= 1ull << (sid & 63);
      }
      alist[g] = g;
    }
  }
  const int64_t wb = v.wn_off[p], nwin = v.wn_off[p + 1] - wb;
  for (int64_t w = 0; w < nwin; w++) {
    const int w0 = v.win_cand[wb + w + p], w1 = v.win_cand[wb + w + p + 1];   // (n_windows + 1) entries per problem
    if (lead) {
      for (int i = w0; i < w1; i++) order[i] = i;
      sort_inplace(order + w0, (int64_t)(w1 - w0), LessSupportDesc{support});   // sorted(key=support, reverse=True) is stable
    }
    for (int oi = w0; oi < w1; oi++) {
      int c = 0, sid = 0, best = -1; double best_dist = INFINITY, alen = 0;
      if (lead) { c = order[oi]; sid = sample[c]; alen = fabs((double)svlen[c]); }
      int na_u = na;
#if defined(__HIP_DEVICE_COMPILE__)
      if (WAVE) na_u = __shfl(na, 0, 64);
#endif
      for (int k = 0; k < na_u; k++) {
        int need = 0, g = 0; double dist = 0;
        unsigned long long pa = 0, pb = 0; long long la = 0, lb = 0;
        if (lead) {
          g = alist[k];
          if (svtype == SNF_BND) {
            dist = fabs(gpos[g] - (double)pos[c]) + fabs(gmate[g] - (double)mpos[c]);
            if (dist < best_dist && dist <= (double)(cfg.cluster_merge_bnd * 2) && gmc[g] == mctg[c]) {
              if (!cfg.combine_separate_intra || !((bits[(int64_t)g * nw + (sid >> 6)] >> (sid & 63)) & 1)) { best = g; best_dist = dist; }
            }
          } else {
            dist = fabs(gpos[g] - (double)pos[c]) + fabs(fabs(glen[g]) - alen);
            const double minlen = fabs(glen[g]) < alen ? fabs(glen[g]) : alen;
            if (minlen > 0 && dist < best_dist && dist <= (double)cfg.combine_match * sqrt(minlen) && dist <= (double)cfg.combine_match_max) {
              if (!(cfg.combine_separate_intra && ((bits[(int64_t)g * nw + (sid >> 6)] >> (sid & 63)) & 1))) {
                if (cfg.combine_pctseq != 0.0) {  // SVGroup.align_call: needs the edit distance
                  need = 1;
                  pa = (unsigned long long)((gsrc[g] ? v.g_alt_pool : v.alt_pool) + galo[g]); la = gahi[g] - galo[g];
                  pb = (unsigned long long)(v.alt_pool + v.alt_off[c0 + c]); lb = v.alt_off[c0 + c + 1] - v.alt_off[c0 + c];
                } else { best = g; best_dist = dist; }
              }
            }
          }
        }
        // SVGroup.align_call accepts iff (len_mean - d) / len_mean > combine_pctseq: only distances up to the largest d that
        // still satisfies it matter, so the alignment is banded with that cut-off (Ukkonen) and answers -1 beyond it
        long long kmax = -1;
        if (need && lead) {
          const double gl = glen[g];
          long long d0 = (long long)floor(gl * (1.0 - cfg.combine_pctseq));
          const long long dcap = la > lb ? la : lb;          // no distance exceeds the longer string
          if (d0 > dcap) d0 = dcap;
          if (d0 < 0) d0 = 0;
          while (d0 < dcap && ((gl - (double)(d0 + 1)) / gl) > cfg.combine_pctseq) d0++;
          while (d0 >= 0 && !(((gl - (double)d0) / gl) > cfg.combine_pctseq)) d0--;
          if (d0 < 0) need = 0;                               // not even identical strings would be accepted
          kmax = d0;
        }
        int64_t d = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        if (WAVE) {
          need = __shfl(need, 0, 64);
          if (need) {
            pa = __shfl(pa, 0, 64); pb = __shfl(pb, 0, 64); la = __shfl(la, 0, 64); lb = __shfl(lb, 0, 64); kmax = __shfl(kmax, 0, 64);
            if (ed_wave_band_fits((int64_t)la, (int64_t)lb, (int64_t)kmax)) d = ed_wave_pair_k_any((const uint8_t*)pa, (int64_t)la, (const uint8_t*)pb, (int64_t)lb, (int64_t)kmax);
            else { d = ed_wave_pair((const uint8_t*)pa, (int64_t)la, (const uint8_t*)pb, (int64_t)lb, carry); if (d > kmax) d = -1; }
          }
        } else
#endif
        if (need) d = ed_serial_k((const uint8_t*)pa, (int64_t)la, (const uint8_t*)pb, (int64_t)lb, (int64_t)kmax, ed_scr);
        if (need && lead) {
          const int sx = (int)(p & 63);
          atomic_add_u64(&v.stats[(0 * 64 + sx) * 16], 1ull);
          atomic_add_u64(&v.stats[(1 * 64 + sx) * 16], (unsigned long long)(la + lb));
          atomic_add_u64(&v.stats[(2 * 64 + sx) * 16], (unsigned long long)la * (unsigned long long)lb);
        }
        if (need && d < 0) need = 0;                          // beyond the cut-off: rejected
        if (need && lead && ((glen[g] - (double)d) / glen[g]) > cfg.combine_pctseq) { best = g; best_dist = dist; }
      }
      if (lead) {
        if (best < 0) {  // SVGroup.from_candidate
          const int g = ng++;
          gpos[g] = (double)pos[c]; glen[g] = fabs((double)svlen[c]); gmate[g] = (double)mpos[c];
          gsize[g] = 1; gmc[g] = mctg[c];
          galo[g] = v.alt_off[c0 + c]; gahi[g] = v.alt_off[c0 + c + 1]; gsrc[g] = 0;
          for (int ww = 0; ww < nw; ww++) bits[(int64_t)g * nw + ww] = 0;
          bits[(int64_t)g * nw + (sid >> 6)] |= 1ull << (sid & 63);
          alist[na++] = g;
          out[c] = g;
        } else {         // SVGroup.add_candidate: multiply, add, append, divide
          const int g = best;
          const double n = (double)gsize[g];
          gpos[g] *= n; glen[g] *= n;
          gpos[g] += (double)pos[c]; glen[g] += fabs((double)svlen[c]);
          if (svtype == SNF_BND) { gmate[g] *= n; gmate[g] += (double)mpos[c]; }
          gsize[g]++;
          const double n1 = (double)gsize[g];
          gpos[g] /= n1; glen[g] /= n1;
          bits[(int64_t)g * nw + (sid >> 6)] |= 1ull << (sid & 63);
          if (svtype == SNF_BND) gmate[g] /= n1;
          out[c] = g;
        }
      }
    }
    if (lead) {  // flush (parallel.py:553-556): groups near the window's last bin stay active for the next window
      const double thr = v.win_thr[wb + w];
      if (thr >= 0) {
        const double bin = (double)v.win_bin[wb + w];
        int nk = 0;
        for (int k = 0; k < na; k++) { const int g = alist[k]; if (fabs(gpos[g] - bin) < thr) alist[nk++] = g; }
        na = nk;
      }
    }
  }
}

SNF_HD void combine_problem_body(int64_t p, const CombineView& v) { combine_run<false>(p, v); }

__global__ void __launch_bounds__(64) combine_problem_wave(const CombineView v, int64_t np) {
  for (int64_t p = blockIdx.x; p < np; p += gridDim.x) combine_run<true>(p, v);
}

}  // namespace snf
using namespace snf;
SNF_KERNEL(combine_problem, CombineView)
SNF_KERNEL(group_call, GroupCallView)

namespace {
DevArena g_combine_arenas[SNF_MAX_DEVICES];
DevArena g_groupcall_arenas[SNF_MAX_DEVICES];

// an array of the call inside the arena: where it lives and (inputs) the host vector that fills it
struct Slot { size_t off, bytes; const void* src; };
}  // namespace

extern "C" int snf_combine_resolve_batch(const snf_config_t* cfg, int device, const snf_combine_problem_t* P, int64_t np) {
  if (np <= 0) return 0;
  if (!cfg || !P) return 1;
  if (device < 0 || device >= SNF_MAX_DEVICES) return 1;
  std::vector<int32_t> svtype(np), ncands(np), ngroups(np), nwords(np);
  std::vector<int64_t> c_off(np + 1, 0), g_off(np + 1, 0), s_off(np + 1, 0), w_off(np + 1, 0), k_off(np + 1, 0), e_off(np + 1, 0);
  for (int64_t p = 0; p < np; p++) {
    svtype[p] = P[p].svtype; ncands[p] = P[p].n_cands; ngroups[p] = P[p].n_groups;
    nwords[p] = (P[p].n_sample_ids + 63) / 64 > 0 ? (P[p].n_sample_ids + 63) / 64 : 1;
    c_off[p + 1] = c_off[p] + P[p].n_cands; g_off[p + 1] = g_off[p] + P[p].n_groups;
    s_off[p + 1] = s_off[p] + P[p].n_cands + P[p].n_groups;
    w_off[p + 1] = w_off[p] + (int64_t)nwords[p] * (P[p].n_cands + P[p].n_groups);
    int64_t maxlen = 1;
    for (int i = 0; i < P[p].n_cands; i++) { int64_t l = P[p].alt_off[i + 1] - P[p].alt_off[i]; if (l > maxlen) maxlen = l; }
    for (int g = 0; g < P[p].n_groups; g++) { int64_t l = P[p].g_alt_off[g + 1] - P[p].g_alt_off[g]; if (l > maxlen) maxlen = l; }
    // per-column carry bytes: only the multi-pass wave form needs them (a band of more than 63 blocks: strings > 4 kb
    // that differ a lot); block states of the serial form: emulation / SNF_COMBINE_THREAD
    k_off[p + 1] = k_off[p] + (maxlen > 4000 ? maxlen + 8 : 8);
    e_off[p + 1] = e_off[p] + ed_serial_scratch_words(maxlen);
  }
  // flush windows per problem (a plain problem = one window that flushes nothing)
  std::vector<int64_t> wn_off(np + 1, 0);
  std::vector<int32_t> win_cand, win_bin; std::vector<double> win_thr;
  for (int64_t p = 0; p < np; p++) {
    const snf_combine_problem_t& q = P[p];
    if (q.n_windows > 0) {
      if (!q.win_off || !q.win_bin || !q.win_thr || q.win_off[0] != 0 || q.win_off[q.n_windows] != q.n_cands) return 1;
      for (int w = 0; w < q.n_windows; w++) {
        if (q.win_off[w + 1] < q.win_off[w]) return 1;
        win_cand.push_back(q.win_off[w]); win_bin.push_back(q.win_bin[w]); win_thr.push_back(q.win_thr[w]);
      }
      win_cand.push_back(q.win_off[q.n_windows]);
      wn_off[p + 1] = wn_off[p] + q.n_windows;
    } else {
      win_cand.push_back(0); win_cand.push_back(q.n_cands); win_bin.push_back(0); win_thr.push_back(-1.0);
      wn_off[p + 1] = wn_off[p] + 1;
    }
  }
  const int64_t NC = c_off[np], NG = g_off[np];
  std::vector<int32_t> pos, svlen, support, sample, mctg, mpos, g_size, g_mctg, g_samples;
  std::vector<double> g_pm, g_lm, g_mm;
  std::vector<int64_t> alt_off(1, 0), g_alt_off(1, 0), g_s_off(1, 0);
  pos.reserve(NC); svlen.reserve(NC); support.reserve(NC); sample.reserve(NC); mctg.reserve(NC); mpos.reserve(NC); alt_off.reserve(NC + 1);
  int64_t alt_bytes = 0, g_alt_bytes = 0;
  for (int64_t p = 0; p < np; p++) {
    const snf_combine_problem_t& q = P[p];
    for (int i = 0; i < q.n_cands; i++) {
      if (q.sample_id[i] < 0 || q.sample_id[i] >= (q.n_sample_ids > 0 ? q.n_sample_ids : 1)) return 1;
      pos.push_back(q.pos[i]); svlen.push_back(q.svlen[i]); support.push_back(q.support[i]); sample.push_back(q.sample_id[i]);
      mctg.push_back(q.mate_contig ? q.mate_contig[i] : 0); mpos.push_back(q.mate_ref_start ? q.mate_ref_start[i] : 0);
      alt_bytes += q.alt_off[i + 1] - q.alt_off[i];
      alt_off.push_back(alt_bytes);
    }
    for (int g = 0; g < q.n_groups; g++) {
      g_pm.push_back(q.g_pos_mean[g]); g_lm.push_back(q.g_len_mean[g]); g_mm.push_back(q.g_mate_mean ? q.g_mate_mean[g] : 0.0);
      g_size.push_back(q.g_size[g]); g_mctg.push_back(q.g_mate_contig ? q.g_mate_contig[g] : 0);
      g_alt_bytes += q.g_alt_off[g + 1] - q.g_alt_off[g];
      g_alt_off.push_back(g_alt_bytes);
      for (int64_t k = q.g_samples_off[g]; k < q.g_samples_off[g + 1]; k++) g_samples.push_back(q.g_samples[k]);
      g_s_off.push_back((int64_t)g_samples.size());
    }
  }
  (void)NG;
  // ---- arena layout: inputs first (one host-to-device copy), state / scratch / output behind
  ArenaLayout L;
  std::vector<Slot> in;
  auto put = [&](const auto& vec) { typedef typename std::decay<decltype(vec)>::type V; typedef typename V::value_type T;
                                    const size_t o = L.add<T>(vec.size() + 2); in.push_back({o, vec.size() * sizeof(T), vec.data()}); return o; };
  const size_t o_svtype = put(svtype), o_ncands = put(ncands), o_ngroups = put(ngroups), o_nwords = put(nwords);
  const size_t o_coff = put(c_off), o_goff = put(g_off), o_soff = put(s_off), o_woff = put(w_off), o_koff = put(k_off), o_eoff = put(e_off);
  const size_t o_pos = put(pos), o_svlen = put(svlen), o_support = put(support), o_sample = put(sample), o_mctg = put(mctg), o_mpos = put(mpos);
  const size_t o_altoff = put(alt_off);
  const size_t o_gpm = put(g_pm), o_glm = put(g_lm), o_gmm = put(g_mm), o_gsize = put(g_size), o_gmctg = put(g_mctg), o_galtoff = put(g_alt_off);
  const size_t o_gsoff = put(g_s_off), o_gsamples = put(g_samples);
  const size_t o_wnoff = put(wn_off), o_wincand = put(win_cand), o_winbin = put(win_bin), o_winthr = put(win_thr);
  const size_t o_alt = L.add<uint8_t>((size_t)alt_bytes + 32), o_galt = L.add<uint8_t>((size_t)g_alt_bytes + 32);   // (Myers reads 8-byte words)
  const size_t in_end = (L.at + 255) & ~(size_t)255;
  const size_t S = (size_t)s_off[np];
  const size_t o_stpos = L.add<double>(S), o_stlen = L.add<double>(S), o_stmate = L.add<double>(S);
  const size_t o_stsize = L.add<int32_t>(S), o_stmctg = L.add<int32_t>(S), o_stalo = L.add<int64_t>(S), o_stahi = L.add<int64_t>(S);
  const size_t o_stsrc = L.add<uint8_t>(S), o_stalist = L.add<int32_t>(S), o_bits = L.add<uint64_t>((size_t)w_off[np]);
  const size_t o_order = L.add<int32_t>((size_t)NC), o_carry = L.add<int8_t>((size_t)k_off[np] + 16);
  const bool thread_form = getenv("SNF_COMBINE_THREAD") != nullptr;
  const size_t o_edscr = L.add<uint64_t>(thread_form ? (size_t)e_off[np] + 16 : 16);
  const size_t o_out = L.add<int32_t>((size_t)NC);
  const size_t o_stats = L.add<unsigned long long>(3 * 64 * 16);
  DevArena& A = g_combine_arenas[device];
  std::lock_guard<std::mutex> hold(A.mu);
  if (!A.ensure(device, L.at)) return 1;
  uint8_t *h = A.h, *d = A.d;
  for (const Slot& sl : in) if (sl.bytes) memcpy(h + sl.off, sl.src, sl.bytes);
  {  // the ALT strings go straight from the caller's pools into the staging mirror
    uint8_t* w = h + o_alt; uint8_t* wg = h + o_galt;
    for (int64_t p = 0; p < np; p++) {
      const snf_combine_problem_t& q = P[p];
      if (q.n_cands > 0) { const int64_t n = q.alt_off[q.n_cands] - q.alt_off[0]; memcpy(w, q.alt_pool + q.alt_off[0], (size_t)n); w += n; }
      if (q.n_groups > 0) { const int64_t n = q.g_alt_off[q.n_groups] - q.g_alt_off[0]; memcpy(wg, q.g_alt_pool + q.g_alt_off[0], (size_t)n); wg += n; }
    }
    memset(w, 0, 32); memset(wg, 0, 32);
  }
  CombineView v{};
  v.cfg = *cfg; v.n_problems = np;
  v.svtype = (const int32_t*)(d + o_svtype); v.n_cands = (const int32_t*)(d + o_ncands); v.n_groups = (const int32_t*)(d + o_ngroups);
  v.n_words = (const int32_t*)(d + o_nwords);
  v.c_off = (const int64_t*)(d + o_coff); v.g_off = (const int64_t*)(d + o_goff); v.s_off = (const int64_t*)(d + o_soff);
  v.w_off = (const int64_t*)(d + o_woff); v.k_off = (const int64_t*)(d + o_koff); v.e_off = (const int64_t*)(d + o_eoff);
  v.pos = (const int32_t*)(d + o_pos); v.svlen = (const int32_t*)(d + o_svlen); v.support = (const int32_t*)(d + o_support);
  v.sample_id = (const int32_t*)(d + o_sample); v.mate_contig = (const int32_t*)(d + o_mctg); v.mate_pos = (const int32_t*)(d + o_mpos);
  v.alt_off = (const int64_t*)(d + o_altoff); v.alt_pool = d + o_alt;
  v.g_pos_mean = (const double*)(d + o_gpm); v.g_len_mean = (const double*)(d + o_glm); v.g_mate_mean = (const double*)(d + o_gmm);
  v.g_size = (const int32_t*)(d + o_gsize); v.g_mate_contig = (const int32_t*)(d + o_gmctg);
  v.g_alt_off = (const int64_t*)(d + o_galtoff); v.g_alt_pool = d + o_galt;
  v.g_samples_off = (const int64_t*)(d + o_gsoff); v.g_samples = (const int32_t*)(d + o_gsamples);
  v.st_pos = (double*)(d + o_stpos); v.st_len = (double*)(d + o_stlen); v.st_mate = (double*)(d + o_stmate);
  v.st_size = (int32_t*)(d + o_stsize); v.st_mctg = (int32_t*)(d + o_stmctg);
  v.st_alt_lo = (int64_t*)(d + o_stalo); v.st_alt_hi = (int64_t*)(d + o_stahi); v.st_alt_src = d + o_stsrc; v.st_alist = (int32_t*)(d + o_stalist);
  v.wn_off = (const int64_t*)(d + o_wnoff); v.win_cand = (const int32_t*)(d + o_wincand); v.win_bin = (const int32_t*)(d + o_winbin);
  v.win_thr = (const double*)(d + o_winthr);
  v.st_bits = (uint64_t*)(d + o_bits); v.order = (int32_t*)(d + o_order); v.carry = (int8_t*)(d + o_carry);
  v.ed_scratch = (uint64_t*)(d + o_edscr);
  v.out_group = (int32_t*)(d + o_out);
  v.stats = (unsigned long long*)(d + o_stats);
  hipStream_t st = A.stream;
  bool ok = hipMemcpyAsync(d, h, in_end, hipMemcpyHostToDevice, st) == hipSuccess;
  ok = ok && hipMemsetAsync(d + o_stats, 0, 3 * 64 * 16 * sizeof(unsigned long long), st) == hipSuccess;
  ok = ok && hipEventRecord(A.ev0, st) == hipSuccess;
  if (ok) {
    if (thread_form) hipLaunchKernelGGL(combine_problem, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, st, v, np);
    else hipLaunchKernelGGL(combine_problem_wave, dim3((unsigned)(np < 65536 ? np : 65536)), dim3(64), 0, st, v, np);
    ok = hipGetLastError() == hipSuccess;
  }
  ok = ok && hipEventRecord(A.ev1, st) == hipSuccess;
  ok = ok && hipMemcpyAsync(h + o_stats, d + o_stats, 3 * 64 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) == hipSuccess;
  ok = ok && (NC == 0 || hipMemcpyAsync(h + o_out, d + o_out, (size_t)NC * sizeof(int32_t), hipMemcpyDeviceToHost, st) == hipSuccess);
  ok = ok && hipStreamSynchronize(st) == hipSuccess;
  if (!ok) return 1;
  { float ms = 0; if (hipEventElapsedTime(&ms, A.ev0, A.ev1) == hipSuccess) A.last_kernel_ms = ms; }
  {
    const unsigned long long* hs = (const unsigned long long*)(h + o_stats);
    for (int c = 0; c < 3; c++) { unsigned long long t = 0; for (int k = 0; k < 64; k++) t += hs[(c * 64 + k) * 16]; A.last_stats[c] = (long long)t; }
    A.last_stats[3] = (long long)in_end;
  }
  const int32_t* h_out = (const int32_t*)(h + o_out);
  for (int64_t p = 0; p < np; p++)
    for (int i = 0; i < P[p].n_cands; i++) P[p].out_group[i] = h_out[(size_t)(c_off[p] + i)];
  return 0;
}

// measurement hook: the kernel time of the last snf_combine_resolve_batch on `device` (HIP events on the arena's stream) and
// what it aligned: stats[0] alignments (SVGroup.align_call evaluations), [1] bytes of the aligned strings, [2] cells of their
// full DP matrices, [3] bytes staged host -> HBM
extern "C" int snf_combine_last_stats(int device, double* kernel_ms, int64_t* stats4) {
  if (device < 0 || device >= SNF_MAX_DEVICES || !kernel_ms || !stats4) return 1;
  DevArena& A = g_combine_arenas[device];
  std::lock_guard<std::mutex> hold(A.mu);
  *kernel_ms = A.last_kernel_ms;
  for (int k = 0; k < 4; k++) stats4[k] = A.last_stats[k];
  return 0;
}

// SVGroup.call + the keep / flush walk for all groups of a merge (include/sniffles_amd.h; body: snf_group_call.h)
extern "C" int snf_combine_call_groups(const snf_group_call_config_t* cfg, int device, int64_t n_groups, const int64_t* group_off,
                                       const int32_t* member, int64_t n_cands, const snf_group_cand_t* cand, const int32_t* cand_win,
                                       const int32_t* group_win_hi, int64_t n_windows, const int32_t* win_bin, const double* win_thr,
                                       snf_group_out_t* out, uint8_t* member_chosen, double* member_pos_mean) {
  if (n_groups <= 0) return 0;
  if (!cfg || !group_off || !member || !cand || !cand_win || !group_win_hi || !win_bin || !win_thr || !out || !member_chosen || !member_pos_mean) return 1;
  if (device < 0 || device >= SNF_MAX_DEVICES || n_cands < 0 || n_windows <= 0) return 1;
  const int64_t NM = group_off[n_groups];
  if (group_off[0] != 0 || NM < 0) return 1;
  for (int64_t g = 0; g < n_groups; g++) {
    if (group_off[g + 1] < group_off[g] || group_win_hi[g] < 1 || group_win_hi[g] > n_windows) return 1;
    int32_t wprev = -1;
    for (int64_t k = group_off[g]; k < group_off[g + 1]; k++) {
      const int32_t c = member[k];
      if (c < 0 || c >= n_cands) return 1;
      const int32_t w = cand_win[c];
      if (w < wprev || w < 0 || w >= group_win_hi[g]) return 1;     // add order follows the windows
      wprev = w;
    }
  }
  ArenaLayout L;
  const size_t o_goff = L.add<int64_t>((size_t)n_groups + 1), o_member = L.add<int32_t>((size_t)NM), o_cand = L.add<snf_group_cand_t>((size_t)n_cands);
  const size_t o_cwin = L.add<int32_t>((size_t)n_cands), o_ghi = L.add<int32_t>((size_t)n_groups);
  const size_t o_wbin = L.add<int32_t>((size_t)n_windows), o_wthr = L.add<double>((size_t)n_windows);
  const size_t in_end = (L.at + 255) & ~(size_t)255;
  const size_t o_out = L.add<snf_group_out_t>((size_t)n_groups), o_chosen = L.add<uint8_t>((size_t)NM), o_pm = L.add<double>((size_t)NM);
  const size_t out_end = (L.at + 255) & ~(size_t)255;
  const size_t o_scr = L.add<int32_t>((size_t)NM);
  DevArena& A = g_groupcall_arenas[device];
  std::lock_guard<std::mutex> hold(A.mu);
  if (!A.ensure(device, L.at)) return 1;
  uint8_t *h = A.h, *d = A.d;
  memcpy(h + o_goff, group_off, ((size_t)n_groups + 1) * sizeof(int64_t));
  memcpy(h + o_member, member, (size_t)NM * sizeof(int32_t));
  memcpy(h + o_cand, cand, (size_t)n_cands * sizeof(snf_group_cand_t));
  memcpy(h + o_cwin, cand_win, (size_t)n_cands * sizeof(int32_t));
  memcpy(h + o_ghi, group_win_hi, (size_t)n_groups * sizeof(int32_t));
  memcpy(h + o_wbin, win_bin, (size_t)n_windows * sizeof(int32_t));
  memcpy(h + o_wthr, win_thr, (size_t)n_windows * sizeof(double));
  GroupCallView v{};
  v.cfg = *cfg; v.n_groups = n_groups;
  v.group_off = (const int64_t*)(d + o_goff); v.member = (const int32_t*)(d + o_member); v.cand = (const snf_group_cand_t*)(d + o_cand);
  v.cand_win = (const int32_t*)(d + o_cwin); v.group_win_hi = (const int32_t*)(d + o_ghi);
  v.win_bin = (const int32_t*)(d + o_wbin); v.win_thr = (const double*)(d + o_wthr);
  v.out = (snf_group_out_t*)(d + o_out); v.chosen = d + o_chosen; v.pos_mean = (double*)(d + o_pm); v.scratch = (int32_t*)(d + o_scr);
  hipStream_t st = A.stream;
  bool ok = hipMemcpyAsync(d, h, in_end, hipMemcpyHostToDevice, st) == hipSuccess;
  ok = ok && hipEventRecord(A.ev0, st) == hipSuccess;
  if (ok) { hipLaunchKernelGGL(group_call, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, v, n_groups); ok = hipGetLastError() == hipSuccess; }
  ok = ok && hipEventRecord(A.ev1, st) == hipSuccess;
  ok = ok && hipMemcpyAsync(h + o_out, d + o_out, out_end - o_out, hipMemcpyDeviceToHost, st) == hipSuccess;
  ok = ok && hipStreamSynchronize(st) == hipSuccess;
  if (!ok) return 1;
  { float ms = 0; if (hipEventElapsedTime(&ms, A.ev0, A.ev1) == hipSuccess) A.last_kernel_ms = ms; }
  memcpy(out, h + o_out, (size_t)n_groups * sizeof(snf_group_out_t));
  memcpy(member_chosen, h + o_chosen, (size_t)NM);
  memcpy(member_pos_mean, h + o_pm, (size_t)NM * sizeof(double));
  for (int64_t g = 0; g < n_groups; g++) if (out[g].emit < 0) return 1;
  return 0;
}
