"""Measurement of the multi-sample merge kernels - dev tool, not the driver's bench.

  python tools/bench_combine.py            # on a GPU box

(1) snf_edit_distance_batch: pairs of INS ALT-like sequences (length mixture of SURVEY.md 8d: 70 % 50+Exp(150),
    20 % N(320,15), 10 % N(6000,100); 4 % substitutions/indels between the two), GPU pairs/s and 64x64-cell word steps/s
    against the exact scalar DP of the oracle on a sample.
(2) snf_combine_resolve_batch: independent flush windows of a synthetic 10-sample population (36 candidates per window
    on average), GPU windows/s against the oracle's serial resolve on the same windows.
Prints one JSON line.
"""
import json
import os, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from sniffles_amd import lib, cluster, sv
from sniffles_amd.config import SnifflesConfig
import oracle

rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

def ins_len():
    u = rng.random()
    if u < 0.7: return int(50 + rng.exponential(150))
    if u < 0.9: return max(50, int(rng.normal(320, 15)))
    return max(50, int(rng.normal(6000, 100)))

def mutate(a, rate=0.04):
    out = []
    for ch in a:
        u = rng.random()
        if u < rate / 3: continue
        if u < 2 * rate / 3: out.append(int(ACGT[rng.integers(4)]))
        out.append(int(ch) if u >= rate else int(ACGT[rng.integers(4)]))
    return np.array(out, dtype=np.uint8)

def main():
    oracle.build()
    res = {}
    # ---- (1) edit distance
    n_pairs = 20000
    pairs = []
    for _ in range(n_pairs):
        a = ACGT[rng.integers(0, 4, ins_len())]
        pairs.append((a.tobytes(), mutate(a).tobytes()))
    lib.edit_distance_batch(pairs[:64])                       # warm-up / module load
    t0 = time.perf_counter(); d = lib.edit_distance_batch(pairs); t_gpu = time.perf_counter() - t0
    cells = sum(len(a) * len(b) for a, b in pairs)
    sample = pairs[:300]
    t0 = time.perf_counter(); dref = [oracle.edit_distance(a, b) for a, b in sample]; t_cpu = time.perf_counter() - t0
    assert list(d[:len(sample)]) == dref
    cells_s = sum(len(a) * len(b) for a, b in sample)
    res["edit_distance"] = dict(pairs=n_pairs, bytes=sum(len(a) + len(b) for a, b in pairs), dp_cells=cells,
                                gpu_s_incl_h2d_d2h=round(t_gpu, 4), gpu_pairs_per_s=round(n_pairs / t_gpu),
                                gpu_word_steps_per_s=round(cells / 64 / t_gpu), cpu_cells_per_s=round(cells_s / t_cpu),
                                gpu_cells_per_s=round(cells / t_gpu), sample_checked=len(sample))
    # ---- (2) resolve_block_groups windows
    cfg = SnifflesConfig(); n_samples = 10
    cfg.snf_input_info = [dict(internal_id=s) for s in range(n_samples)]; cfg.mode = "combine"
    windows = []
    n_windows = 4000
    for w in range(n_windows):
        cands = []; base = 100000 + 5000 * w
        for site in range(6):
            pos = base + 700 * site + int(rng.integers(0, 300)); L = ins_len(); allele = ACGT[rng.integers(0, 4, L)]
            for s in range(n_samples):
                if rng.random() > 0.6: continue
                c = sv.new_call()
                c.svtype = "INS"; c.pos = pos + int(rng.integers(-5, 6)); c.svlen = max(45, int(L * (1 + rng.normal(0, 0.015))))
                c.support = int(rng.integers(3, 30)); c.sample_internal_id = s
                c.alt = mutate(allele, 0.03).tobytes().decode("latin-1"); c.end = c.pos
                cands.append(c)
        windows.append(("INS", cands, []))
    n_c = sum(len(w[1]) for w in windows)
    cluster.resolve_block_groups_batch([("INS", list(c), []) for _, c, _ in windows[:8]], cfg)   # warm-up
    t0 = time.perf_counter(); groups = cluster.resolve_block_groups_batch(windows, cfg); t_gpu = time.perf_counter() - t0
    keep2 = []; packed = [cluster.pack_problem(svt, c, [], keep2) for svt, c, _ in windows]
    t0 = time.perf_counter(); lib.combine_resolve_batch(cfg, [q for q, _ in packed]); t_call = time.perf_counter() - t0
    # parity of a sample of the windows against the oracle's serial resolve (same packed problems)
    keep = []; n_chk = int(os.environ.get("SNF_BENCH_ORACLE_WINDOWS", "150")); t_or = 0.0
    for svt, cands, _ in windows[:n_chk]:
        q, out_gpu = cluster.pack_problem(svt, cands, [], keep)
        lib.combine_resolve_batch(cfg, [q])
        got = list(out_gpu)
        q2, out_or = cluster.pack_problem(svt, cands, [], keep)
        t1 = time.perf_counter(); oracle.combine_resolve(cfg, q2); t_or += time.perf_counter() - t1
        assert got == list(out_or)
    res["resolve_block_groups"] = dict(oracle_windows_checked=n_chk, oracle_windows_per_s=round(n_chk / t_or, 1),
windows=n_windows, candidates=n_c, gpu_s_incl_pack_and_replay=round(t_gpu, 3), c_abi_call_s_incl_h2d_kernel_d2h=round(t_call, 4),
                                       windows_per_s=round(n_windows / t_gpu), candidates_per_s=round(n_c / t_gpu),
                                       groups=sum(len(g) for g in groups))
    # ---- (3) the same windows as CHAINS (CombineTask.execute: kept groups seed the next window): 40 chains x 100 windows
    chains = []
    for k in range(40):
        cands, woff, wbin, wthr = [], [0], [], []
        for svt, c, _ in windows[k * 100:(k + 1) * 100]:
            cands.extend(c); woff.append(len(cands)); wbin.append(max(x.pos for x in c) // 100 * 100); wthr.append(2500.0)
        chains.append(("INS", cands, woff, wbin, wthr))
    keep3 = []; packed3 = [cluster.pack_problem(t, c, [], keep3, (wo, wb, wt)) for t, c, wo, wb, wt in chains]
    lib.combine_resolve_batch(cfg, [q for q, _ in packed3[:2]])
    t0 = time.perf_counter(); lib.combine_resolve_batch(cfg, [q for q, _ in packed3]); t_ch = time.perf_counter() - t0
    whole = [out for _, out in packed3]
    # the product path (cluster.resolve_chains_batch): chains cut where no candidate can reach an earlier group
    n_sub = sum(len(cluster.chain_cuts(t, c, wo, cfg)) - 1 for t, c, wo, _, _ in chains)
    t0 = time.perf_counter(); cut = cluster.resolve_chains_batch(chains, cfg); t_cut = time.perf_counter() - t0
    assert all(np.array_equal(a[:len(c[1])], b[:len(c[1])]) for a, b, c in zip(whole, cut, chains))
    res["resolve_chains"] = dict(chains=len(chains), windows=100 * len(chains), candidates=sum(len(c[1]) for c in chains),
                                 whole_chains_c_abi_call_s=round(t_ch, 4), whole_chains_windows_per_s=round(100 * len(chains) / t_ch),
                                 sub_chains=n_sub, cut_s_incl_pack=round(t_cut, 4), windows_per_s=round(100 * len(chains) / t_cut),
                                 identical_assignment=True)
    print(json.dumps(res))

if __name__ == "__main__":
    main()
