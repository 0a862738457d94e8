#!/bin/bash
O=gpurun_out/r02bc; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5"
for rep in a b c; do
for z in 0 1; do
if [ $z = 1 ]; then export SNF_CONS_ONE_STREAM=1; else unset SNF_CONS_ONE_STREAM; fi
timeout 300 python bench.py $Q > $O/c1_$z$rep.json 2> $O/c1_$z$rep.err
python - <<PY
import json
d=json.load(open('$O/c1_$z$rep.json')); print('one_stream $z', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:3]])
PY
done
done
