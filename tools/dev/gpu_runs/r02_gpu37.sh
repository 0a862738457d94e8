#!/bin/bash
bash tools/r02_profile.sh r02fin2 > gpurun_out/profile_fin2.log 2>&1
head -12 gpurun_out/prof_r02fin2/kernel_stats_3_in_flight.csv
python - <<PY
import json
d=json.load(open('gpurun_out/prof_r02fin2/bench_under_rocprof.json')); r=d['roofline']; print('under rocprof:', round(d['ms_per_step'],3), r['kernel'], r['kernel_ms'], [(k['name'],k['ms']) for k in r['top_kernels'][:5]])
PY
