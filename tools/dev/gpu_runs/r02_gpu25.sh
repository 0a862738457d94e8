#!/bin/bash
O=gpurun_out/r02ae; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=12
Q="--no-cpu-baseline --no-wall-clock --steps 10 --warmup 3 --inflight 1"
for e in 0 1; do
  for c in 1 2 3; do
  if [ $e = 1 ]; then export SNF_DBG_E1=1; else unset SNF_DBG_E1; fi
  SNF_SERIAL=1 timeout 300 python bench.py --config $c $Q > $O/s_${e}_$c.json 2> $O/s_${e}_$c.err
  python - <<PY
import json
d=json.load(open('$O/s_${e}_$c.json')); t={k['name']:k['ms'] for k in d['roofline']['top_kernels']}; print('ablate=$e config=$c e1w', t.get('e1w_finalize'), 'x_big_finalize', t.get('x_big_finalize'))
PY
  done
done
