#!/bin/bash
O=gpurun_out/r02bd; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c1.json 2> $O/c1.err
SNF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-wall-clock > $O/c1_dist.json 2> $O/c1_dist.err
timeout 900 python bench.py --config 4 > $O/c4.json 2> $O/c4.err
for c in c1 c1_dist c4; do python - <<PY
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,3), round(d['ms_per_step'],3), d.get('verified'), d['config'].get('host_binding'), d['config'].get('parallelism'), (d.get('wall_clock') or {}).get('batched'), d['config'].get('host_phases_ms'))
PY
done
