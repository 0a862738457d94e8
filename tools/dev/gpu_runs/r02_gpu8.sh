#!/bin/bash
set -x
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-wall-clock > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wall-clock > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
for c in 1 2; do python -c "
import json
d=json.load(open('$O/bench_c$c.json')); print($c, round(d['value']/1e6,1), round(d['ms_per_step'],3), [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])"; done
