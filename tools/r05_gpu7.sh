#!/bin/bash
# round 5, seventh GPU session: what the driver runs at round end - the GPU suite, smoke(), the default line - on the final tree,
# and the per-task upload split behind the cached genotype table
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu7.log 2>&1; tail -3 gpurun_out/pytest_gpu7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_7.json 2> gpurun_out/bench_default_7.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_7.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"), "one in flight", d["config"]["ms_per_pass_one_batch_in_flight"])
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")), wc["batched"]["end_to_end_ms"])
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "seconds", "error", "reference_error")})
print("config 4 baseline", json.dumps(d["configs"]["4"].get("cpu_baseline"))[:900])
r = d["roofline"]
print("roofline", r["kernel"], r["kernel_ms"], r["frac"], r.get("rocprof_ms"), r.get("rocprof_frac"), "| stage", r.get("dominant_stage"))
print("issue", r.get("issue"))
PY
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -4
