#!/bin/bash
# round 5, fifth GPU session: the whole GPU suite, the default line as the driver runs it (all legs), the upload split per task
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; tail -3 gpurun_out/pytest_gpu5.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_5.json 2> gpurun_out/bench_default_5.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_5.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"), "one in flight", d["config"]["ms_per_pass_one_batch_in_flight"])
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
for k, v in (wc.get("worker_processes") or {}).items():
    print("workers", k, v if not isinstance(v, dict) else {x: v[x] for x in ("hot_all_ms", "ingest_all_ms", "hw_queues_per_process", "n_out")})
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "records_compared", "vs_reference_all_cores", "seconds", "error", "reference_error")})
r = d["roofline"]
print("roofline", r["kernel"], r["kernel_ms"], r["frac"], "| stage", r.get("dominant_stage"))
print([(k["name"], k["ms"]) for k in r["top_kernels"]])
PY
tail -3 gpurun_out/bench_default_5.err
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -5
