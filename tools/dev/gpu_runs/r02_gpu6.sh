#!/bin/bash
set -x
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
grep "SNF_PROF\] upload\|read index" $O/bench_c1.err | tail -4
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
timeout 600 python bench.py --config 3 --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo rc=$? >> $O/bench_c3.err
timeout 300 python bench.py --config 0 --steps 20 --warmup 3 > $O/bench_c0.json 2> $O/bench_c0.err; echo rc=$? >> $O/bench_c0.err
for c in 1 2 3 0; do python -c "
import json
d=json.load(open('$O/bench_c$c.json')); print($c, round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), d['wall_clock']['batched'], d['wall_clock']['per_task_api'], round(d['cpu_baseline']['value']/1e6,1))"; done
