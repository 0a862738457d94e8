"""Per-kernel SQ counters of a pass (tools/sq_all.sh): duration, VALU floor, resident waves, instruction mix."""
import csv, glob, sys, collections, re
def short(n):
    n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n); return n.replace("snf::", "")[:44]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k in acc:
    c = {x: y / max(1, len(nl[k][x])) for x, y in acc[k].items()}
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else 0.0
    valu_us = c.get("SQ_INSTS_VALU", 0) * 4 / 1024 / 2.4e3     # 4 cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz
    wc = c.get("SQ_WAVE_CYCLES", 0)
    rows.append((d, k, valu_us, c, wc))
rows.sort(reverse=True)
# LDS: SQ_LDS_BANK_CONFLICT = cycles the LDS pipe spent on bank conflicts, SQ_LDS_IDX_ACTIVE = cycles it was busy with indexed accesses:
# their ratio is the share of the LDS pipe's time that conflicts cost (0 where a kernel has no LDS traffic)
print("%-44s %8s %8s %9s %9s %9s %9s %9s %9s %11s %11s %8s" % ("kernel (per launch)", "us", "VALU us", "waves", "VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "LDS_BANK_CF", "LDS_IDX_ACT", "cf/act"))
for d, k, vu, c, wc in rows:
    bc, ia = c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-44s %8.1f %8.1f %9.0f %9.3g %9.3g %9.3g %9.3g %9.3g %11.3g %11.3g %8.3f  wave_cycles %.3g busy %.3g" % (k, d, vu, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0),
          c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_VMEM_WR", 0), bc, ia, (bc / ia if ia else 0.0), wc, c.get("SQ_BUSY_CYCLES", 0)))
