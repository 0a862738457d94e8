#!/bin/bash
O=gpurun_out/r02an; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=12
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
for e in 5; do
SNF_OCC_E1=$e SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/serial_e$e.json 2> $O/serial_e$e.err
SNF_OCC_E1=$e timeout 300 python bench.py $Q > $O/c1_e$e.json 2> $O/c1_e$e.err
for c in serial_e$e c1_e$e; do python - <<PY
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['roofline']['gpu_ms_all_kernels'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])
PY
done
done
