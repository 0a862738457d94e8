#!/bin/bash
set -x
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
timeout 900 python bench.py --config 4 > $O/bench_c4.json 2> $O/bench_c4.err; echo rc=$? >> $O/bench_c4.err
tail -3 $O/bench_c4.err
python -c "
import json
d=json.load(open('$O/bench_c1.json')); print(round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], d.get('verified'), d['wall_clock'])
d=json.load(open('$O/bench_c4.json')); print(d['value'], d['ms_per_step'], d['config']['rank0'], d.get('verified'), d['cpu_baseline']['value'])"
