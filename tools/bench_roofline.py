"""bench.py: what the roofline block of the line is assembled from - the committed rocprofv3 summaries of this round (quoted only for
the kernel sources they were measured on) and the measured PCIe rate of the box."""
from __future__ import annotations

import json
import os

from tools.bench_common import EMU, ROOT

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# rocprofv3 summaries of this round, collected with tools/profile.sh.  They are only quoted when they were measured on the
# kernels this run executes: profiles/r06_profile_meta.json records the hash of sniffles_amd/csrc they belong to.
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
ROCPROF_STATS = os.path.join(ROOT, "profiles", "r06_kernel_stats_default.csv")
PROFILE_META = os.path.join(ROOT, "profiles", "r06_profile_meta.json")
ROCPROF_NAMES = {"e45w_consensus_large": "e45w_consensus<2,", "e45w_consensus_small": "e45w_consensus<1,", "d2w_call": "d2w_call<",
                 "e1w_finalize": "e1w_finalize<",
                 "d1w_refine": "d1w_refine", "d4_coverage": "d4_coverage", "a4_binstats": "a4k_binstats", "a6_scatter": "a6k_scatter",
                 "e4c_copy": "e4c_copy",
                 "a1_keys": "a1_keys", "a0_keep": "a0k_keep", "c1_mergeruns": "c1_mergeruns", "d3_rnames": "d3rk_rnames",
                 "b1_seedmetrics": "b1k_seedmetrics",
                 "d5w_covsum": "d5w_covsum", "f4_emit": "f4w_emit", "f5_alt": "f5w_alt", "f3_rank": "f3k_rank", "w1_hist": "w1_hist",
                 "w3_scatter": "w3_scatter",
                 "w4s_segment": "w4s_segment<", "w6t_emit": "w6t_emit", "d2g_call8": "d2g_call<8", "d1g_refine8": "d1g_refine<8"}
SQ_FILE = os.path.join(ROOT, "profiles", "r06_sq_all.txt")


def committed_profiles():
    """(rocprofv3 average ms per kernel name, PMC traffic per kernel, note) of the committed round-5 profiles - or empty dicts and
    the reason when they belong to other kernel sources than the ones built here."""
    try:
        from sniffles_amd import build
        meta = json.load(open(PROFILE_META))
        if meta.get("csrc_sha") != build._lib_digest():
            return {}, {}, "profiles/r06_* were collected on other kernel sources (stale): not quoted"
        import csv
        avg = {}
        for r in csv.DictReader(open(ROCPROF_STATS)):
            for short, pat in ROCPROF_NAMES.items():
                if pat in r["Name"].replace("snf::", "").replace(" ", "").replace("void", "") or pat in r["Name"]:
                    avg.setdefault(short, float(r["AverageNs"]) / 1e6)
        pmc = json.load(open(PMC_FILE))["kernels"] if os.path.exists(PMC_FILE) else {}
        return avg, pmc, ("rocprofv3 --kernel-trace --stats of the default command, profiles/r06_kernel_stats_default.csv "
                          "(same kernel sources: hash checked)")
    except Exception as e:  # noqa: BLE001
        return {}, {}, f"no committed profile for these sources ({type(e).__name__})"


def committed_issue():
    """{kernel name: dict(valu_us, kernel_us, frac, lds_conflict_share)} from profiles/r06_sq_all.txt (tools/sq_all.sh: SQ_INSTS_VALU x
    4 cycles on the 1024 SIMDs at 2.4 GHz against the kernel's duration, one batch in flight; SQ_LDS_BANK_CONFLICT /
    SQ_LDS_IDX_ACTIVE) - the ISSUE roof of the kernels that are bound by instruction issue rather than by bytes.
    Empty when the file belongs to other kernel sources."""
    try:
        from sniffles_amd import build
        if json.load(open(PROFILE_META)).get("csrc_sha") != build._lib_digest():
            return {}
        out = {}
        for ln in open(SQ_FILE):
            f = ln.split()
            if len(f) < 12 or f[0] in ("kernel", "#"):
                continue
            # the name may hold spaces (template arguments): the numeric columns are the last ones
            try:
                k = ln[:44].strip()
                nums = ln[44:].split()
                us, valu_us = float(nums[0]), float(nums[1])
                conflict = float(nums[10]) if len(nums) > 10 else None
            except (ValueError, IndexError):
                continue
            for short, pat in ROCPROF_NAMES.items():
                if pat.replace(" ", "") in k.replace(" ", ""):
                    out.setdefault(short, dict(valu_us=valu_us, kernel_us=us, frac=round(valu_us / us, 3) if us > 0 else None,
                                               lds_bank_conflict_share_of_lds_cycles=conflict))
        return out
    except Exception:  # noqa: BLE001
        return {}


def pcie_d2h_peak_gbs(torch, nbytes=32 << 20, reps=8):
    """Device -> pinned host copy rate of this box (GB/s): the roof of the result path, measured, not assumed."""
    if EMU:
        return float("nan")
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    for _ in range(2):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
