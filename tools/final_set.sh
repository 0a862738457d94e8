#!/bin/bash
# tools/final_set.sh [TAG]: the bench lines that are kept for judging, one box -> gpurun_out/final_TAG/ (copy into profiles/ as rNN_final_bench_*.json)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final_$TAG; rm -rf $O; mkdir -p $O
last() { grep '^{"metric"' | tail -1; }
python bench.py 2> $O/default.err | last > $O/default.json                                         # exactly what the driver runs (N = 1)
Q="--no-configs --no-cpu-baseline --no-wall-clock"
python bench.py --output candidates $Q 2> /dev/null | last > $O/candidates.json
python bench.py --inflight 1 $Q 2> /dev/null | last > $O/one_in_flight.json
SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2> /dev/null | last > $O/shared_1rank.json
SNF_BENCH_FORCE_DIST=1 SNF_BENCH_GATHER=rccl python bench.py --gpus 1 $Q 2> /dev/null | last > $O/rccl_1rank.json
SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --scaling strong $Q 2> /dev/null | last > $O/strong_1rank.json
python bench.py --config 4 2> /dev/null | last > $O/config4.json
python bench.py --genomes 2 $Q 2> /dev/null | last > $O/genomes2.json
python bench.py --genomes 4 $Q 2> /dev/null | last > $O/genomes4.json
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read())
        r = d.get("roofline", {})
        print(os.path.basename(f), "ms_per_step", d.get("ms_per_step"), "value %.4g" % d.get("value", 0), d.get("unit"), "| verified", d.get("verified"), d.get("verified_vs_reference"),
              "| roofline", r.get("kernel"), r.get("kernel_ms"), r.get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
