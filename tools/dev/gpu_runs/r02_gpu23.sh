#!/bin/bash
O=gpurun_out/r02al; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=12
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/serial.json 2> $O/serial.err
timeout 600 python bench.py --no-wall-clock --steps 20 --warmup 5 > $O/c1.json 2> $O/c1.err
timeout 300 python bench.py $Q > $O/c1b.json 2> $O/c1b.err
for c in serial c1 c1b; do python - <<PY
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), d['roofline']['gpu_ms_all_kernels'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])
PY
done
for c in 2 3; do
timeout 900 python bench.py --config $c --no-wall-clock > $O/cfg$c.json 2> $O/cfg$c.err
python - <<PY
import json
d=json.load(open('$O/cfg$c.json')); print('config $c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:6]])
PY
done
