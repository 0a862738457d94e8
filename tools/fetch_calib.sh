#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts per access pattern (tools/probe/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $O/known.json
for C in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- /tmp/fetch_calib > $O/$C.log 2>&1; done
python - <<PY > $R/gpurun_out/fetch_calib.txt
import csv, glob, json, collections
known = json.load(open("$O/known.json"))
print(json.dumps(known))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % C, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == C]
    by = collections.defaultdict(float)
    for r in rows: by[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
    for (d, k), v in sorted(by.items()): print("%s dispatch %2d %-12s %10.1f MiB counted (KiB units -> bytes %.0f)" % (C, d, k, v / 1024, v * 1024))
PY
cat $R/gpurun_out/fetch_calib.txt; rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
