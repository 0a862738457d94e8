"""`resolve_block_groups` with the reference's signature (reference `src/sniffles/cluster.py:356-390`).

The single-sample clustering entry points of the reference's `cluster.py` (`resolve`, `merge_inner`, `resplit`,
`resplit_bnd`) have no Python counterpart here: they run inside `Task.call_candidates` on the GPU
(`sniffles_amd/parallel.py`)."""
from __future__ import annotations

from . import abi, lib
from .soa import SVT
from .sv import SVGroup


def _alt_bytes(alt) -> bytes:
    return alt.encode("latin-1") if isinstance(alt, str) else bytes(alt)


def pack_problem(svtype, svcands, groups_initial, keep, windows=None):
    contig_ids = {}

    def cid(name):
        return contig_ids.setdefault(name, len(contig_ids))

    bnd = svtype == "BND"
    cands = dict(pos=[c.pos for c in svcands], svlen=[c.svlen for c in svcands], support=[c.support for c in svcands],
                 sample_id=[c.sample_internal_id for c in svcands],
                 mate_contig=[cid(c.bnd_info.mate_contig) if bnd else 0 for c in svcands],
                 mate_ref_start=[c.bnd_info.mate_ref_start if bnd else 0 for c in svcands],
                 alts=[_alt_bytes(c.alt) for c in svcands])
    groups = dict(pos_mean=[g.pos_mean for g in groups_initial], len_mean=[g.len_mean for g in groups_initial],
                  mate_mean=[float(g.bnd_mate_ref_start_mean) if bnd else 0.0 for g in groups_initial],
                  size=[len(g.candidates) for g in groups_initial],
                  mate_contig=[cid(g.bnd_mate_contig) if bnd else 0 for g in groups_initial],
                  alts=[_alt_bytes(g.candidates[0].alt) for g in groups_initial],
                  samples=[g.included_samples for g in groups_initial])
    ids = [c.sample_internal_id for c in svcands] + [s for g in groups_initial for s in g.included_samples]
    n_ids = (max(ids) + 1) if ids else 1
    return abi.combine_problem(SVT[svtype], cands, groups, n_ids, keep, windows)


def apply_assignment(svcands, groups_initial, out_group):
    """Replay SVGroup.from_candidate / add_candidate in the reference's order (support descending, stable) with the
    group index the GPU chose for every candidate; the running means are recomputed by the same float operations."""
    groups = groups_initial
    n0 = len(groups_initial)
    created = {}
    order = sorted(range(len(svcands)), key=lambda i: svcands[i].support, reverse=True)
    for i in order:
        g = int(out_group[i])
        if g < n0:
            groups[g].add_candidate(svcands[i])
        elif g in created:
            created[g].add_candidate(svcands[i])
        else:
            created[g] = SVGroup.from_candidate(svcands[i])
            groups.append(created[g])
    return groups


def resolve_block_groups(svtype, svcands, groups_initial, config, device: int = 0, _lib=None):
    """For clustering groups of SVs when combining .snf files (multi-call).  Mutates and returns `groups_initial`."""
    keep = []
    q, out = pack_problem(svtype, svcands, groups_initial, keep)
    lib.combine_resolve_batch(config, [q], device=device, _lib=_lib)
    return apply_assignment(svcands, groups_initial, out)


def resolve_block_groups_batch(problems, config, device: int = 0, _lib=None):
    """Several independent (svtype, svcands, groups_initial) problems in one launch - e.g. the flush windows of
    different contigs / SV types of CombineTask.execute (parallel.py:524-534).  Returns the list of group lists."""
    keep, packed = [], []
    for svtype, svcands, groups_initial in problems:
        packed.append(pack_problem(svtype, svcands, groups_initial, keep))
    lib.combine_resolve_batch(config, [q for q, _ in packed], device=device, _lib=_lib)
    return [apply_assignment(sc, gi, out) for (_, sc, gi), (_, out) in zip(problems, packed)]


def chain_cuts(svtype, svcands, win_off, config):
    """Window indices at which a chain of flush windows falls apart into independent sub-chains.

    A candidate joins a group only if `abs(group.pos_mean - cand.pos)` (plus a non-negative length or mate term) is at
    most `combine_match_max` (`cluster_merge_bnd * 2` for BND) - cluster.py:369-379.  A group's `pos_mean` is the mean
    of positions of earlier candidates, so when every candidate from window w on lies further than that gate (+1 bp for
    the rounding of the running mean) to the right of every earlier candidate, no group that exists before w can
    ever receive another candidate: the groups kept across the cut are carried and flushed by the host replay alone
    (CombineTask.execute) and the windows from w on are a chain of their own.  Whole-genome merges fall apart into
    thousands of short chains this way, which is what fills the device; the greedy inside a sub-chain is unchanged."""
    nw = len(win_off) - 1
    if nw <= 1:
        return [0, nw]
    gate = float(config.cluster_merge_bnd) * 2 if svtype == "BND" else float(config.combine_match_max)
    gate = max(gate, 0.0) + 1.0
    lo = [min((c.pos for c in svcands[win_off[w]:win_off[w + 1]]), default=float("inf")) for w in range(nw)]
    hi = [max((c.pos for c in svcands[win_off[w]:win_off[w + 1]]), default=float("-inf")) for w in range(nw)]
    for w in range(nw - 2, -1, -1):      # suffix minimum / prefix maximum: no assumption about the order of the windows
        lo[w] = min(lo[w], lo[w + 1])
    for w in range(1, nw):
        hi[w] = max(hi[w], hi[w - 1])
    return [0] + [w for w in range(1, nw) if lo[w] - hi[w - 1] > gate] + [nw]


def resolve_chains_batch(chains, config, device: int = 0, _lib=None, cut: bool = None):
    """Whole chains of flush windows in one launch.  chains: list of (svtype, svcands, win_off, win_bin, win_thr) with
    svcands the concatenation of the windows' candidates; after window w the groups with
    abs(pos_mean - win_bin[w]) < win_thr[w] stay active for window w+1 (CombineTask.execute, parallel.py:553-556).
    Returns per chain the group number of every candidate (new groups numbered in creation order over the chain).
    Chains are cut where no candidate can reach an earlier group (`chain_cuts`; SNF_COMBINE_NO_CUT=1 / cut=False keeps
    them whole): every sub-chain is one work item of the kernel, the group numbers are made chain-wide here.
    A whole-genome merge is tens of thousands of sub-chains, so they are packed as rows of one table over shared
    candidate columns (`abi.combine_chain_problems`), not as one ctypes structure each."""
    import os

    import numpy as np
    if cut is None:
        cut = os.environ.get("SNF_COMBINE_NO_CUT", "0") != "1"
    cols = {k: [] for k in ("pos", "svlen", "support", "sample_id", "mate_contig", "mate_ref_start")}
    alts, wstart, wbin, wthr = [], [], [], []
    codes, c_lo, c_hi, w_lo, w_hi, chain_of = [], [], [], [], [], []
    span = []                                    # per chain: its candidate range in the shared numbering
    for ci, (svtype, svcands, win_off, win_bin, win_thr) in enumerate(chains):
        c0, w0 = len(alts), len(wbin)
        bnd = svtype == "BND"
        cols["pos"].extend(c.pos for c in svcands)
        cols["svlen"].extend(c.svlen for c in svcands)
        cols["support"].extend(c.support for c in svcands)
        cols["sample_id"].extend(c.sample_internal_id for c in svcands)
        if bnd:
            ids = {}
            cols["mate_contig"].extend(ids.setdefault(c.bnd_info.mate_contig, len(ids)) for c in svcands)
            cols["mate_ref_start"].extend(c.bnd_info.mate_ref_start for c in svcands)
        else:
            cols["mate_contig"].extend([0] * len(svcands))
            cols["mate_ref_start"].extend([0] * len(svcands))
        alts.extend(_alt_bytes(c.alt) for c in svcands)
        wstart.extend(c0 + o for o in win_off[:-1])
        wbin.extend(win_bin)
        wthr.extend(win_thr)
        span.append((c0, len(alts)))
        cuts = chain_cuts(svtype, svcands, win_off, config) if cut and len(win_bin) else [0, len(win_bin)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                codes.append(SVT[svtype]); chain_of.append(ci)
                c_lo.append(c0 + win_off[a]); c_hi.append(c0 + win_off[b]); w_lo.append(w0 + a); w_hi.append(w0 + b)
    n_c = len(alts)
    if not codes:
        return [np.full(max(hi - lo, 1), -1, np.int32) for lo, hi in span]
    keep = []
    n_ids = max(cols["sample_id"]) + 1 if n_c else 1
    arr, out = abi.combine_chain_problems(codes, c_lo, c_hi, w_lo, w_hi, cols, alts, wstart + [n_c], wbin, wthr, n_ids, keep)
    lib.combine_resolve_batch(config, arr, device=device, _lib=_lib)
    # chain-wide group numbers: sub-chain s of a chain starts after the groups its predecessors created
    lo, hi = np.asarray(c_lo, np.int64), np.asarray(c_hi, np.int64)
    created = np.array([int(out[a:b].max()) + 1 if b > a else 0 for a, b in zip(c_lo, c_hi)], np.int64)
    ch = np.asarray(chain_of, np.int64)
    cum = np.cumsum(created) - created                                # exclusive over all sub-chains ...
    first = np.r_[True, ch[1:] != ch[:-1]]
    base = cum - np.maximum.accumulate(np.where(first, cum, 0))       # ... relative to the chain's first sub-chain
    delta = np.zeros(n_c + 1, np.int64)                               # sub-chains are disjoint candidate ranges
    np.add.at(delta, lo, base)
    np.add.at(delta, hi, -base)
    out[:n_c] += np.cumsum(delta[:n_c]).astype(np.int32)
    return [out[a:b] if b > a else np.full(1, -1, np.int32) for a, b in span]
