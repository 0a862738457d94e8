#!/bin/bash
O=gpurun_out/r02ai; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5"
for rep in a b c d; do
  for w in 2 3; do
  timeout 300 python bench.py $Q --inflight $w > $O/w${w}$rep.json 2> $O/w${w}$rep.err
  python - <<PY
import json
d=json.load(open('$O/w${w}$rep.json')); print('inflight $w', round(d['value']/1e6,1), round(d['ms_per_step'],3))
PY
  done
done
