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
  // flush windows: wn_off[p] = first window of problem p; win_cand has (windows + 1) entries per problem, stored at
  // index (wn_off[p] + p + w); win_bin / win_thr per window (thr < 0: nothing is flushed)
  const int64_t* wn_off; const int32_t* win_cand; const int32_t* win_bin; const double* win_thr;
  int32_t* out_group;
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
#if !defined(SNF_EMU) && defined(__HIP_DEVICE_COMPILE__)
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
#if !defined(SNF_EMU) && defined(__HIP_DEVICE_COMPILE__)
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
        int64_t d = 0;
#if !defined(SNF_EMU) && defined(__HIP_DEVICE_COMPILE__)
        if (WAVE) {
          need = __shfl(need, 0, 64);
          if (need) {
            pa = __shfl(pa, 0, 64); pb = __shfl(pb, 0, 64); la = __shfl(la, 0, 64); lb = __shfl(lb, 0, 64);
            d = ed_wave_pair((const uint8_t*)pa, (int64_t)la, (const uint8_t*)pb, (int64_t)lb, carry);
          }
        } else
#endif
        if (need) d = ed_serial((const uint8_t*)pa, (int64_t)la, (const uint8_t*)pb, (int64_t)lb, carry);
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

#ifndef SNF_EMU
__global__ void __launch_bounds__(64) combine_problem_wave(const CombineView v, int64_t np) {
  for (int64_t p = blockIdx.x; p < np; p += gridDim.x) combine_run<true>(p, v);
}
#endif

}  // namespace snf
using namespace snf;
SNF_KERNEL(combine_problem, CombineView)

namespace {
template <class T>
T* dev_up(const std::vector<T>& h, std::vector<void*>& frees, bool& ok, size_t extra = 0) {
#ifndef SNF_EMU
  void* p = nullptr;
  size_t bytes = (h.size() + extra + 2) * sizeof(T);
  if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
  frees.push_back(p);
  if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
  return (T*)p;
#else
  (void)ok;
  T* p = (T*)malloc((h.size() + extra + 2) * sizeof(T));
  frees.push_back(p);
  if (!h.empty()) memcpy(p, h.data(), h.size() * sizeof(T));
  return p;
#endif
}
template <class T>
T* dev_scratch(size_t n, std::vector<void*>& frees, bool& ok) { std::vector<T> e; return dev_up<T>(e, frees, ok, n); }
}  // namespace

extern "C" int snf_combine_resolve_batch(const snf_config_t* cfg, int device, const snf_combine_problem_t* P, int64_t np) {
  if (np <= 0) return 0;
  if (!cfg || !P) return 1;
#ifndef SNF_EMU
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device < 0 || device >= nd) return 1;
  if (hipSetDevice(device) != hipSuccess) return 1;
#else
  (void)device;
#endif
  std::vector<int32_t> svtype(np), ncands(np), ngroups(np), nwords(np);
  std::vector<int64_t> c_off(np + 1, 0), g_off(np + 1, 0), s_off(np + 1, 0), w_off(np + 1, 0), k_off(np + 1, 0);
  for (int64_t p = 0; p < np; p++) {
    svtype[p] = P[p].svtype; ncands[p] = P[p].n_cands; ngroups[p] = P[p].n_groups;
    nwords[p] = (P[p].n_sample_ids + 63) / 64 > 0 ? (P[p].n_sample_ids + 63) / 64 : 1;
    c_off[p + 1] = c_off[p] + P[p].n_cands; g_off[p + 1] = g_off[p] + P[p].n_groups;
    s_off[p + 1] = s_off[p] + P[p].n_cands + P[p].n_groups;
    w_off[p + 1] = w_off[p] + (int64_t)nwords[p] * (P[p].n_cands + P[p].n_groups);
    int64_t maxlen = 1;
    for (int i = 0; i < P[p].n_cands; i++) { int64_t l = P[p].alt_off[i + 1] - P[p].alt_off[i]; if (l > maxlen) maxlen = l; }
    for (int g = 0; g < P[p].n_groups; g++) { int64_t l = P[p].g_alt_off[g + 1] - P[p].g_alt_off[g]; if (l > maxlen) maxlen = l; }
    k_off[p + 1] = k_off[p] + maxlen + 8;
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
  std::vector<uint8_t> alt_pool, g_alt_pool;
  for (int64_t p = 0; p < np; p++) {
    const snf_combine_problem_t& q = P[p];
    for (int i = 0; i < q.n_cands; i++) {
      if (q.sample_id[i] < 0 || q.sample_id[i] >= (q.n_sample_ids > 0 ? q.n_sample_ids : 1)) return 1;
      pos.push_back(q.pos[i]); svlen.push_back(q.svlen[i]); support.push_back(q.support[i]); sample.push_back(q.sample_id[i]);
      mctg.push_back(q.mate_contig ? q.mate_contig[i] : 0); mpos.push_back(q.mate_ref_start ? q.mate_ref_start[i] : 0);
      alt_pool.insert(alt_pool.end(), q.alt_pool + q.alt_off[i], q.alt_pool + q.alt_off[i + 1]);
      alt_off.push_back((int64_t)alt_pool.size());
    }
    for (int g = 0; g < q.n_groups; g++) {
      g_pm.push_back(q.g_pos_mean[g]); g_lm.push_back(q.g_len_mean[g]); g_mm.push_back(q.g_mate_mean ? q.g_mate_mean[g] : 0.0);
      g_size.push_back(q.g_size[g]); g_mctg.push_back(q.g_mate_contig ? q.g_mate_contig[g] : 0);
      g_alt_pool.insert(g_alt_pool.end(), q.g_alt_pool + q.g_alt_off[g], q.g_alt_pool + q.g_alt_off[g + 1]);
      g_alt_off.push_back((int64_t)g_alt_pool.size());
      for (int64_t k = q.g_samples_off[g]; k < q.g_samples_off[g + 1]; k++) g_samples.push_back(q.g_samples[k]);
      g_s_off.push_back((int64_t)g_samples.size());
    }
  }
  std::vector<void*> frees;
  bool ok = true;
  CombineView v{};
  v.cfg = *cfg; v.n_problems = np;
  v.svtype = dev_up(svtype, frees, ok); v.n_cands = dev_up(ncands, frees, ok); v.n_groups = dev_up(ngroups, frees, ok);
  v.n_words = dev_up(nwords, frees, ok);
  v.c_off = dev_up(c_off, frees, ok); v.g_off = dev_up(g_off, frees, ok); v.s_off = dev_up(s_off, frees, ok);
  v.w_off = dev_up(w_off, frees, ok); v.k_off = dev_up(k_off, frees, ok);
  v.pos = dev_up(pos, frees, ok); v.svlen = dev_up(svlen, frees, ok); v.support = dev_up(support, frees, ok);
  v.sample_id = dev_up(sample, frees, ok); v.mate_contig = dev_up(mctg, frees, ok); v.mate_pos = dev_up(mpos, frees, ok);
  v.alt_off = dev_up(alt_off, frees, ok); v.alt_pool = dev_up(alt_pool, frees, ok, 16);
  v.g_pos_mean = dev_up(g_pm, frees, ok); v.g_len_mean = dev_up(g_lm, frees, ok); v.g_mate_mean = dev_up(g_mm, frees, ok);
  v.g_size = dev_up(g_size, frees, ok); v.g_mate_contig = dev_up(g_mctg, frees, ok);
  v.g_alt_off = dev_up(g_alt_off, frees, ok); v.g_alt_pool = dev_up(g_alt_pool, frees, ok, 16);
  v.g_samples_off = dev_up(g_s_off, frees, ok); v.g_samples = dev_up(g_samples, frees, ok);
  const size_t S = (size_t)s_off[np];
  v.st_pos = dev_scratch<double>(S, frees, ok); v.st_len = dev_scratch<double>(S, frees, ok); v.st_mate = dev_scratch<double>(S, frees, ok);
  v.st_size = dev_scratch<int32_t>(S, frees, ok); v.st_mctg = dev_scratch<int32_t>(S, frees, ok);
  v.st_alt_lo = dev_scratch<int64_t>(S, frees, ok); v.st_alt_hi = dev_scratch<int64_t>(S, frees, ok);
  v.st_alt_src = dev_scratch<uint8_t>(S, frees, ok); v.st_alist = dev_scratch<int32_t>(S, frees, ok);
  v.wn_off = dev_up(wn_off, frees, ok); v.win_cand = dev_up(win_cand, frees, ok); v.win_bin = dev_up(win_bin, frees, ok);
  v.win_thr = dev_up(win_thr, frees, ok);
  v.st_bits = dev_scratch<uint64_t>((size_t)w_off[np], frees, ok);
  v.order = dev_scratch<int32_t>((size_t)NC, frees, ok);
  v.carry = dev_scratch<int8_t>((size_t)k_off[np], frees, ok);
  int32_t* d_out = dev_scratch<int32_t>((size_t)NC, frees, ok);
  v.out_group = d_out;
  (void)NG;
  std::vector<int32_t> h_out((size_t)NC + 1);
  if (ok) {
#ifndef SNF_EMU
    if (getenv("SNF_COMBINE_THREAD")) hipLaunchKernelGGL(combine_problem, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, 0, v, np);
    else hipLaunchKernelGGL(combine_problem_wave, dim3((unsigned)(np < 65536 ? np : 65536)), dim3(64), 0, 0, v, np);
    if (hipDeviceSynchronize() != hipSuccess) ok = false;
    if (ok && NC && hipMemcpy(h_out.data(), d_out, (size_t)NC * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
#else
    combine_problem(v, np);
    if (NC) memcpy(h_out.data(), d_out, (size_t)NC * sizeof(int32_t));
#endif
  }
  for (void* p : frees) {
#ifndef SNF_EMU
    (void)hipFree(p);
#else
    free(p);
#endif
  }
  if (!ok) return 1;
  for (int64_t p = 0; p < np; p++)
    for (int i = 0; i < P[p].n_cands; i++) P[p].out_group[i] = h_out[(size_t)(c_off[p] + i)];
  return 0;
}
