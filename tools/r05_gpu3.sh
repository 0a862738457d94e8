#!/bin/bash
# round 5, third GPU session: the staged result path as the library's rule (another pass in flight -> through HBM), the GPU tests that
# changed, a same-box A/B, the whole default bench line with the new legs, phase stamps of d1w_refine
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_output_modes.py tests/test_insitu_seam.py tests/test_dropin_api.py tests/test_prefilter.py -m gpu -x -q > gpurun_out/pytest_gpu3.log 2>&1; tail -3 gpurun_out/pytest_gpu3.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: direct:SNF_STAGE_OUT=0 order0:SNF_CONS_ORDER=0 2>&1 | tee gpurun_out/ab_r05_3.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_3.json 2> gpurun_out/bench_default_3.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_3.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"))
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
print("workers", json.dumps(wc.get("worker_processes")))
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "records_compared", "vs_reference_all_cores", "reference_leg_s", "seconds", "error", "reference_error")})
print("roofline", d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
print([(k["name"], k["ms"]) for k in d["roofline"]["top_kernels"]])
PY
tail -5 gpurun_out/bench_default_3.err
SNF_LIB_SO=$R/variants/prof.so SNF_PROF=1 timeout 300 python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 3 --warmup 1 2>&1 | grep "CONS_PROFILE\] d1w" | tail -6
