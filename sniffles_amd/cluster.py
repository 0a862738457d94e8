"""`resolve_block_groups` with the reference's signature (reference `src/sniffles/cluster.py:356-390`) and the view side of
seam B3: `resolve(svtype, lead_provider, config, tr)` yields the clusters of a task as the reference's `cluster.resolve`
does (`cluster.py:219-353`), read back from the GPU after `Task.call_candidates` - the clustering itself (`merge_inner`,
`resplit`, `resplit_bnd`, the merge scan) runs inside the candidate stage on the device, not here.  `dump_clusters_bed`
writes the `--dev-dump-clusters` file of `cluster.py:316-324`."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

from . import abi, lib
from .soa import SOURCES, SVT, SVTYPES
from .sv import SVGroup


@dataclass
class ClusterLead:
    """What a cluster view shows of a `Lead` (leadprov.py:34-56): the fields the dump and the parity diffs read.  `svlen` is
    the value the lead carries at that stage (summed over the fused leads after merge_inner); `row` is its row in the
    task's input table (`TaskInput.leads`)."""
    row: int
    read_qname: str
    ref_start: int
    svlen: int
    source: str


@dataclass
class Cluster:
    id: str
    svtype: str
    contig: str
    start: int
    end: int
    seed: int
    leads: list
    repeat: bool
    leads_long: Optional[int]     # len(cluster.leads_long) of the reference (INS), None where the reference has None

    @property
    def span(self) -> Optional[int]:
        return None if self.end is None or self.start is None else self.end - self.start


def _task_of(lead_provider):
    batch = getattr(lead_provider, "device_batch", None)
    ti = getattr(lead_provider, "ti", None) or getattr(lead_provider, "task_input", None)
    if batch is None or ti is None:
        raise RuntimeError("cluster.resolve needs the task's device batch: call Task.call_candidates first (the clusters are "
                           "built on the GPU; there is no CPU fallback)")
    return batch, ti


def clusters_of(batch, ti, stage: int, task_index: int = 0, provider_start: int = 0) -> dict:
    """{svtype: [Cluster]} of one task at `stage` (0 seeds, 1 merged = the dump point of the reference, 2 = what resolve
    yields), ids as the reference forms them (`CL.{svtype}.{contig}.{start}.{seed_index}` + the resplit suffixes)."""
    c = batch.fetch_clusters(stage)
    L = ti.leads
    out = {t: [] for t in SVTYPES}
    binsize_re = None
    sel = [i for i in range(len(c["svtype"])) if int(c["task_index"][i]) == task_index]
    made = []
    for i in sel:
        svtype = SVTYPES[int(c["svtype"][i])]
        lo, hi = int(c["lead_off"][i]), int(c["lead_off"][i + 1])
        leads = [ClusterLead(row=int(r), read_qname=ti.qname(int(L["qname_id"][r])), ref_start=int(L["ref_start"][r]), svlen=int(s),
                             source=SOURCES[int(L["source"][r])]) for r, s in zip(c["lead"][lo:hi].tolist(), c["lead_svlen"][lo:hi].tolist())]
        cl = Cluster(id=f"CL.{svtype}.{ti.contig}.{provider_start}.{int(c['seed_index'][i])}", svtype=svtype, contig=ti.contig,
                     start=int(c["start"][i]), end=int(c["end"][i]), seed=int(c["seed"][i]), leads=leads, repeat=bool(c["repeat"][i]),
                     leads_long=int(c["n_leads_long"][i]) if svtype == "INS" else None)
        made.append(cl)
        out[svtype].append(cl)
    return out


def _resplit_suffixes(clusters, config):
    """The id suffixes resplit / resplit_bnd append (cluster.py:191, 231-246), recovered from the yielded lead lists: the
    svlen bin that survived the bin merge is the bin of the cluster's first lead; a BND cluster is named after the bin that
    closed it (the first bin of the next cluster of the same (mate contig, is_first) group, or the group's last bin - 0 when
    the group has a single bin)."""
    rb = config.cluster_resplit_binsize
    by_parent = {}
    for cl in clusters:
        by_parent.setdefault(cl.id, []).append(cl)
    for parent, group in by_parent.items():
        svtype = group[0].svtype
        if svtype == "BND":
            if config.dev_no_resplit or (len(group) == 1 and len(group[0].leads) <= 1):
                continue
            thr = config.cluster_merge_bnd

            def key(cl):
                return cl._ident
            for k, cl in enumerate(group):
                nxt = next((g for g in group[k + 1:] if g._ident == cl._ident), None)
                bins = [int(p / thr) * thr if thr > 0 else 0 for p in cl._mate_pos]
                if nxt is not None:
                    nb = [int(p / thr) * thr if thr > 0 else 0 for p in nxt._mate_pos]
                    pos_bin = nb[0]
                else:
                    first_of_ident = next(g for g in group if g._ident == cl._ident)
                    single = first_of_ident is cl and len(set(bins)) == 1
                    pos_bin = 0 if single else bins[-1]
                cl.id = f"{parent}.CHR2.{cl._ident[0]}.POS2.{pos_bin}"
        elif not (config.dev_no_resplit_repeat or config.dev_no_resplit):
            for cl in group:
                cl.id = f"{parent}.{int(abs(cl.leads[0].svlen) / rb) * rb}"


def resolve(svtype, lead_provider, config, tr=None, task_index: int = 0):
    """`cluster.resolve` of the reference for a task whose candidate stage has run on the GPU: yields the refined clusters of
    `svtype` in the reference's order.  `tr` is accepted for the signature's sake (the tandem repeats went in with the task)."""
    batch, ti = _task_of(lead_provider)
    start = getattr(lead_provider, "start", None) or 0
    cls = clusters_of(batch, ti, 2, task_index, start)[svtype]
    if svtype == "BND":
        L = ti.leads
        for cl in cls:
            cl._ident = None
            if cl.leads:
                r0 = cl.leads[0].row
                cl._ident = (ti.contig_name(int(L["mate_contig"][r0])), bool(L["bnd_is_first"][r0]))
            cl._mate_pos = [int(L["mate_ref_start"][ld.row]) for ld in cl.leads]
    _resplit_suffixes(cls, config)
    for cl in cls:
        yield cl


def dump_clusters_bed(lead_provider, config, svtype, handle=None, task_index: int = 0) -> str:
    """The `--dev-dump-clusters` file of one SV type (cluster.py:316-324): the clusters after the merge scan, one BED line
    each with their leads.  Returns the text; writes it to `handle` when given."""
    batch, ti = _task_of(lead_provider)
    start = getattr(lead_provider, "start", None) or 0
    lines = []
    for c in clusters_of(batch, ti, 1, task_index, start)[svtype]:
        info = f"ID={c.id}, #LEADS={len(c.leads)}; "
        for ld in c.leads:
            info += f"(ref_start={ld.ref_start},svlen={ld.svlen},source={ld.source}); "
        lines.append(f"{c.contig}\t{c.start}\t{c.end}\t\"{info}\"\n")
    text = "".join(lines)
    if handle is not None:
        handle.write(text)
    return text


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


def resolve_block_groups(svtype, svcands, groups_initial, config, device: int = 0):
    """For clustering groups of SVs when combining .snf files (multi-call).  Mutates and returns `groups_initial`."""
    keep = []
    q, out = pack_problem(svtype, svcands, groups_initial, keep)
    lib.combine_resolve_batch(config, [q], device=device)
    return apply_assignment(svcands, groups_initial, out)


def resolve_block_groups_batch(problems, config, device: int = 0):
    """Several independent (svtype, svcands, groups_initial) problems in one launch - e.g. the flush windows of
    different contigs / SV types of CombineTask.execute (parallel.py:524-534).  Returns the list of group lists."""
    keep, packed = [], []
    for svtype, svcands, groups_initial in problems:
        packed.append(pack_problem(svtype, svcands, groups_initial, keep))
    lib.combine_resolve_batch(config, [q for q, _ in packed], device=device)
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


def resolve_chains_batch(chains, config, device: int = 0, cut: bool = None):
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
    lib.combine_resolve_batch(config, arr, device=device)
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
