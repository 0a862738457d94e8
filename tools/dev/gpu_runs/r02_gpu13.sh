#!/bin/bash
set -x
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock"
timeout 600 python bench.py --steps 20 --warmup 5 $Q > $O/bench_c1.json 2> $O/bench_c1.err
SNF_SERIAL=1 timeout 300 python bench.py --steps 20 --warmup 5 $Q --inflight 1 > $O/v_serial.json 2> $O/v_serial.err
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 $Q > $O/bench_c2.json 2> $O/bench_c2.err
for c in bench_c1 v_serial bench_c2; do python -c "
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])"; done
bash tools/r02_profile.sh r02m > $O/profile.log 2>&1
