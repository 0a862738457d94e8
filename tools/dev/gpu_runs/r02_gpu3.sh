#!/bin/bash
set -x
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-wall-clock > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
SNF_TIMELINE=1 timeout 300 python bench.py --config 2 --no-cpu-baseline --no-wall-clock --steps 1 --warmup 1 --inflight 1 > $O/c2_timeline.json 2> $O/c2_timeline.err
SNF_TIMELINE=1 timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --steps 1 --warmup 1 --inflight 1 > $O/c1_timeline.json 2> $O/c1_timeline.err
for f in $O/bench_c1.json $O/bench_c2.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], d['config']['ms_per_step_with_read_index_rebuilt_every_pass'], [(k['name'],k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels'][:8]], d.get('verified'), d.get('wall_clock'))"; done
bash tools/r02_profile.sh r02c > $O/profile.log 2>&1
tail -60 $O/profile.log
