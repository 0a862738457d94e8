"""Target SVs for the force-calling tests (GenotypeTask.execute): derived deterministically from the reference's candidate
records of a case, so that the golden generator (reference objects) and the tests (this package's objects) build the
same list.  Near hits around the merge gates, type mismatches, bin-edge positions, BND targets with matching and foreign
mates, far misses, positions near the contig ends (coverage samples outside the vector keep their value)."""
import numpy as np


def target_specs(cand_records, contig_len, seed):
    rng = np.random.default_rng(seed)
    out = []
    k = 0

    def add(svtype, pos, svlen, bnd=None, cov=(0, 0, 0, 0, 0)):
        nonlocal k
        out.append(dict(id=f"T{k}", svtype=svtype, pos=int(max(0, min(contig_len - 1, pos))), svlen=int(svlen), bnd=bnd, cov=list(cov)))
        k += 1
    for c in cand_records:
        if c["svtype"].startswith("SINGLE"):
            continue
        r = rng.random()
        if c["svtype"] == "BND":
            mate = c["bnd"][0] if r < 0.6 else "chrOther"
            add("BND", c["pos"] + int(rng.choice([0, 3, -400, 999, 1000, 1001, -1700])), 0, [mate, int(c["bnd"][1]), bool(c["bnd"][2]), bool(c["bnd"][3])])
            continue
        dpos = int(rng.choice([0, 1, -7, 40, -120, 300, 900, 1100]))
        f = float(rng.choice([1.0, 1.0, 1.02, 0.9, 1.3, 0.5]))
        svlen = int(round(c["svlen"] * f)) or 1
        svtype = c["svtype"] if r < 0.85 else ("DEL" if c["svtype"] == "INS" else "INS")
        add(svtype, c["pos"] + dpos, svlen if svtype != "DEL" else -abs(svlen))
        if r < 0.15:                      # a second target competing for the same candidate
            add(c["svtype"], c["pos"] + dpos + int(rng.integers(-30, 31)), c["svlen"])
    for _ in range(12):                   # misses, bin edges, contig ends, unsupported type, pre-set coverage fields
        add(str(rng.choice(["INS", "DEL", "DUP", "INV"])), int(rng.integers(0, contig_len)), int(rng.integers(50, 4000)))
    add("DEL", 5000 * 7 + 499, -300)
    add("DEL", 5000 * 9 - 499, -300)
    add("INS", 3, 120, cov=(7, 7, 7, 7, 7))
    add("DEL", contig_len - 40, -2000, cov=(9, 8, 7, 6, 5))
    add("CNV", contig_len // 2, 700)
    rng.shuffle(out)
    if out and out[0]["svtype"] == "BND":     # a BND first would be the UnboundLocalError case: covered separately
        j = next(i for i, t in enumerate(out) if t["svtype"] != "BND")
        out[0], out[j] = out[j], out[0]
    return out


def make_targets(specs, svcall_cls, bnd_cls, new_call):
    calls = []
    for t in specs:
        c = new_call(svcall_cls)
        c.id, c.svtype, c.pos, c.svlen, c.contig = t["id"], t["svtype"], t["pos"], t["svlen"], "x"
        c.end = t["pos"] + abs(t["svlen"])
        if t["bnd"] is not None:
            c.bnd_info = bnd_cls(*t["bnd"])
        (c.coverage_upstream, c.coverage_start, c.coverage_center, c.coverage_end, c.coverage_downstream) = t["cov"]
        calls.append(c)
    return calls


def result_records(targets):
    return [dict(id=t.id, match=None if t.genotype_match_sv is None else t.genotype_match_sv.id,
                 dist=None if t.genotype_match_sv is None else float(t.genotype_match_dist),
                 cov=[t.coverage_upstream, t.coverage_start, t.coverage_center, t.coverage_end, t.coverage_downstream],
                 gt=[list(x) if isinstance(x, tuple) else x for x in t.genotypes[0]]) for t in targets]


CASES = ["chr20_30x_ont", "bnd_stale_end", "fuzz_4_2", "chr21_30x_mosaic", "long_ins"]
