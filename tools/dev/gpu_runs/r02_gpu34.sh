#!/bin/bash
O=gpurun_out/r02au; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=8
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
for m in 1 2 4; do
SNF_GRID_MULT=$m SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/s_$m.json 2> $O/s_$m.err
SNF_GRID_MULT=$m timeout 300 python bench.py $Q > $O/c_$m.json 2> $O/c_$m.err
python - <<PY
import json
d=json.load(open('$O/s_$m.json')); t={k['name']:k['ms'] for k in d['roofline']['top_kernels']}
e=json.load(open('$O/c_$m.json'))
print('mult $m d1w', t.get('d1w_refine'), 'd2w', t.get('d2w_call'), 'step', round(e['ms_per_step'],3))
PY
done
