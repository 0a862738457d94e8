"""Canonical records of SNF blocks, shared by the golden generator (reference objects) and the tests (this package's
objects): every attribute an SVCall carries through the SNF container, numpy scalars reduced to Python numbers."""
import hashlib


def _num(x):
    if x is None or isinstance(x, (str, bool)):
        return x
    if hasattr(x, "item"):
        x = x.item()
    return x


def cand_record(c) -> dict:
    d = dict(c.__dict__)
    d.pop("is_single_break", None)          # functools.cached_property residue of the reference class
    fds = d.pop("forward_difference_sampler")
    bi = d.pop("bnd_info")
    rec = {k: _num(v) for k, v in d.items() if k not in ("info", "genotypes", "rnames", "postprocess", "svlens")}
    rec["info"] = {k: _num(v) for k, v in d["info"].items()}
    rec["genotypes"] = {str(k): [_num(v[0]), _num(v[1]), _num(v[2]), _num(v[3]), _num(v[4]), list(v[5])] for k, v in d["genotypes"].items()}
    rec["rnames"] = None if d["rnames"] is None else sorted(d["rnames"])
    rec["postprocess"] = None if d["postprocess"] is None else "set"
    rec["svlens"] = d["svlens"]
    rec["fds"] = [_num(fds.n), _num(fds.m1), _num(fds.m2), _num(fds.last)]
    rec["bnd_info"] = None if bi is None else [bi.mate_contig, _num(bi.mate_ref_start), bool(bi.is_first), bool(bi.is_reverse)]
    return rec


def block_record(block: dict, svtypes) -> dict:
    return dict(cands={t: [cand_record(c) for c in block[t]] for t in svtypes},
                coverage={str(k): _num(v) for k, v in sorted(block["_COVERAGE"].items())})


def file_record(f, contig: str, svtypes) -> dict:
    """f: an opened SNF reader (reference SNFile or sniffles_amd.snf.SNFile) after read_header()."""
    idx = f.index.get(contig, {})
    blocks = {}
    for b in sorted(idx, key=int):
        parts = f.read_blocks(contig, int(b))
        blocks[str(b)] = [block_record(p, svtypes) for p in parts]
    h = f.header
    return dict(blocks=blocks, snf_candidate_count=h["snf_candidate_count"], contigs=sorted(f.index),
                contig_coverages=h["config"].get("contig_coverages"), snf_block_size=h["config"]["snf_block_size"],
                snf_format_version=h["config"]["snf_format_version"], build=h["config"].get("build"))


def sha(path: str) -> str:
    return hashlib.sha256(open(path, "rb").read()).hexdigest()
