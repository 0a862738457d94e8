#!/bin/bash
set -x
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
SNF_NO_BIG_STAGE=1 timeout 600 python -m pytest tests/test_clusters.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_nostage.log 2>&1; echo "rc=$?" >> $O/pytest_nostage.log
tail -3 $O/pytest_nostage.log
timeout 600 python -m pytest tests/test_clusters.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_stage.log 2>&1; echo "rc=$?" >> $O/pytest_stage.log
tail -3 $O/pytest_stage.log
SNF_NO_BIG_STAGE=1 timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-wall-clock > $O/bench_c2_nostage.json 2> $O/bench_c2_nostage.err; echo rc=$? >> $O/bench_c2_nostage.err
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-wall-clock > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
for c in c2_nostage c2; do python -c "
import json
d=json.load(open('$O/bench_$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])"; done
tail -3 $O/bench_c2.err
