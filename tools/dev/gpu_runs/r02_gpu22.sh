#!/bin/bash
O=gpurun_out/r02ac; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=12
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py --config 2 --no-wall-clock > $O/c2.json 2> $O/c2.err
timeout 900 python bench.py --config 3 --no-wall-clock > $O/c3.json 2> $O/c3.err
timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --steps 20 --warmup 5 > $O/c1.json 2> $O/c1.err
for c in c2 c3 c1; do python - <<PY
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:10]])
PY
done
