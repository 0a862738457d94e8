"""Per-kernel SQ counter sums (per launch) from rocprofv3 --pmc passes (dev tool)."""
import csv, glob, sys, collections, re
def short(n):
    n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n); n = n.replace("snf::", "")
    if "e45w_consensus" in n:
        m = re.search(r"e45w_consensus<(\d)", n)
        n = {"1": "e45w_small", "2": "e45w_large", "4": "e45w_rows"}.get(m.group(1) if m else "", n)
    return n[:24]
want = ("e45w", "d1w_refine", "d2w_call", "e1w_finalize", "e4c_copy", "d4_coverage", "c1_mergeruns", "a6k_scatter", "x_big", "x_wave", "x_nmsum")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(want): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[(k, d)].add(r["Dispatch_Id"])
for k in acc:
    n = max(len(v) for (kk, d), v in launches.items() if kk == k)
    c = {x: y / n for x, y in acc[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"== {k}  launches {n}")
    print("   waves %.0f  wave_cycles %.3g  busy_cycles %.3g" % (c.get("SQ_WAVES", 0), wc, c.get("SQ_BUSY_CYCLES", 0)))
    for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS"):
        if x in c: print("   %-22s %.3g  (%.1f %% of wave cycles)" % (x, c[x], 100 * c[x] / wc))
    for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if x in c: print("   %-22s %.3g" % (x, c[x]))
