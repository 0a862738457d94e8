"""Device occupancy of a bench run with several batches in flight, from a rocprofv3 --kernel-trace CSV (tools, not product).

usage: python tools/busy.py <kernel_trace.csv> [passes_to_skip]
Over the passes after the warm-up (a pass = one z0_init launch): wall time per pass, the time at least one kernel ran, the
time exactly one / two / more kernels ran, and per kernel the mean duration and the share of the summed kernel time.
"""
import csv, sys, re, collections

def short(n):
    n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n); return n.replace("snf::", "")[:48]

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
z0 = [e[0] for e in ev if "z0_init" in e[2]]
if len(z0) < skip + 4: sys.exit("not enough passes")
lo, hi = z0[skip], z0[-2]
n_pass = len(z0) - 2 - skip
ev = [e for e in ev if e[0] >= lo and e[0] < hi]
pts = []
for s, e, _ in ev: pts.append((s, 1)); pts.append((min(e, hi), -1))
pts.sort()
depth = 0; last = lo; hist = collections.Counter()
for t, d in pts:
    hist[depth] += t - last; last = t; depth += d
hist[depth] += hi - last
wall = hi - lo
print(f"passes {n_pass}, wall per pass {wall / n_pass / 1e3:.1f} us")
for k in sorted(hist): print(f"  {k} kernels running: {hist[k] / n_pass / 1e3:8.1f} us per pass ({100.0 * hist[k] / wall:.1f} %)")
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in ev: tot[short(n)] += e - s; cnt[short(n)] += 1
allk = sum(tot.values())
print(f"summed kernel time per pass {allk / n_pass / 1e3:.1f} us")
for n, t in tot.most_common(24): print(f"  {t / n_pass / 1e3:8.1f} us per pass  mean {t / cnt[n] / 1e3:7.1f}  x{cnt[n] / n_pass:.1f}  {n}")
