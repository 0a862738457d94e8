"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per-launch averages.

gfx950 correction (MI355X_MICROARCH.md, HBM section; calibrated on a1_keys whose traffic is known exactly):
FETCH_SIZE under-counts by 2x (128-B requests counted as 64 B), WRITE_SIZE is exact; both are in KiB.
"""
import csv, glob, json, re, sys, collections

def short(n):
    n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n); n = n.replace("snf::", "")
    if "rocprim" in n:
        m = re.search(r"(onesweep_iteration|onesweep_histograms|radix_sort\w*|scan_impl|lookback_scan\w*|scan\w*)", n)
        n = "rocprim:" + (m.group(1) if m else n[:40])
    if "e45w_consensus" in n:
        m = re.search(r"e45w_consensus<(\d)", n)
        n = {"1": "e45w_consensus_small", "2": "e45w_consensus_large", "4": "e45w_consensus_rows"}.get(m.group(1) if m else "", n)
    return n

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter: continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}

fe = load(sys.argv[1], "FETCH_SIZE"); wr = load(sys.argv[2], "WRITE_SIZE")
out = {"_about": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --inflight 1 "
                 "--steps 2 --warmup 1 --no-cpu-baseline` on MI355X (30x WG workload, BASELINE configs[1]). Per-launch averages, KiB. "
                 "hbm_bytes = (2*FETCH + WRITE) * 1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B (x2, calibrated on a1_keys: "
                 "known reads 9 B/lead, known writes 13 B/lead), WRITE_SIZE is exact.  hbm_bytes is therefore an UPPER bound: "
                 "tools/probe/fetch_calib.hip (profiles/r04_fetch_calib.txt) shows that FETCH_SIZE counts requests x 64 B - a coalesced "
                 "stream (128-B requests) is reported at half, a gather of 64-byte records (64-B requests: w6t_emit, d1g / d1w_refine, the call "
                 "kernels, d4_coverage) exactly; hbm_bytes_lo = (FETCH + WRITE) * 1024 is the lower bound.", "kernels": {}}
for k in sorted(fe, key=lambda k: -(2 * fe[k][0] + wr.get(k, (0, 1))[0])):
    n = fe[k][1]; f = fe[k][0] / n; w = wr.get(k, (0.0, n))[0] / max(1, wr.get(k, (0, n))[1])
    out["kernels"][k] = dict(launches=n, fetch_kib=round(f, 1), write_kib=round(w, 1), hbm_bytes=int((2 * f + w) * 1024), hbm_bytes_lo=int((f + w) * 1024))
print(json.dumps(out, indent=1))
