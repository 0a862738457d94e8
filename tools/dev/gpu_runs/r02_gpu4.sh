#!/bin/bash
set -x
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
timeout 900 python bench.py --config 4 > $O/bench_c4.json 2> $O/bench_c4.err; echo rc=$? >> $O/bench_c4.err
tail -5 $O/bench_c4.err
timeout 300 python tools/bench_combine.py > $O/bench_combine.json 2> $O/bench_combine.err
cat $O/bench_combine.json | head -c 3000
for f in $O/bench_c1.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], d['config']['ms_per_step_with_read_index_rebuilt_every_pass'], [(k['name'],k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels'][:8]], d.get('verified'))"; done
head -c 4000 $O/bench_c4.json
