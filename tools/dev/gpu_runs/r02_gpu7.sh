#!/bin/bash
set -x
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
SNF_TIMELINE=1 timeout 300 python bench.py --config 2 --no-cpu-baseline --no-wall-clock --steps 1 --warmup 1 --inflight 1 > $O/c2_timeline.json 2> $O/c2_timeline.err
for c in 1 2; do python -c "
import json
d=json.load(open('$O/bench_c$c.json')); print($c, round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), d['wall_clock']['batched'], d['wall_clock']['per_task_api'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:6]])"; done
grep TIMELINE $O/c2_timeline.err | tail -70 | sort -k2 -n | awk '$3>150 {printf "%9.1f %8.1f %s\n",$2,$3,$4}'
