"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV (tools, not product).

usage: python tools/timeline.py <kernel_trace.csv> [step_marker_kernel]
Prints every kernel launch of the last complete step: start offset (us), duration (us), stream/queue, name;
then the union busy time and the idle gaps on the whole device.
"""
import csv, sys, re

def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("snf::", "")
    if "rocprim" in n:
        m = re.search(r"(radix_sort\w*|onesweep\w*|scan\w*|histogram\w*|lookback\w*)", n)
        n = "rocprim:" + (m.group(1) if m else n[:40])
    return n[:60]

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "z0_init"     # the first launch of a pass
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if marker in e[3]]
if len(starts) < 3:
    sys.exit("not enough steps in trace")
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0 = step[0][0]
print(f"# step: {len(step)} launches, span {(ev[b][0]-t0)/1e3:.1f} us")
busy_end = t0; gaps = 0
for s, e, q, n in step:
    gap = s - busy_end
    flag = f"  <-- idle {gap/1e3:.1f}" if gap > 3000 else ""
    if gap > 0: gaps += gap
    busy_end = max(busy_end, e)
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  q{q:>3}  {short(n)}{flag}")
print(f"# device idle inside step: {gaps/1e3:.1f} us")
