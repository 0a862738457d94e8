#!/bin/bash
O=gpurun_out/r02ar; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=14
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
for pf in 0 1; do
if [ $pf = 0 ]; then export SNF_NO_PREFILTER=1; else unset SNF_NO_PREFILTER; fi
SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/serial_$pf.json 2> $O/serial_$pf.err
timeout 300 python bench.py $Q > $O/c1_$pf.json 2> $O/c1_$pf.err
for c in serial_$pf c1_$pf; do python - <<PY
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['roofline']['gpu_ms_all_kernels'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:14]])
PY
done
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-wall-clock > $O/c1v.json 2> $O/c1v.err
python - <<PY
import json
d=json.load(open('$O/c1v.json')); print('verified', d.get('verified'), round(d['ms_per_step'],3))
PY
