#!/bin/bash
O=gpurun_out/r02at; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=8
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
for eb in 8 4 2; do
for c in 1 3; do
SNF_E1_BATCH=$eb SNF_SERIAL=1 timeout 300 python bench.py --config $c $Q --inflight 1 > $O/s_${eb}_$c.json 2> $O/s_${eb}_$c.err
python - <<PY
import json
d=json.load(open('$O/s_${eb}_$c.json')); t={k['name']:k['ms'] for k in d['roofline']['top_kernels']}; print('batch $eb config $c e1w', t.get('e1w_finalize'))
PY
done
done
