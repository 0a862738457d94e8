#!/bin/bash
O=gpurun_out/r02bb; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5"
lscpu | grep -i "numa\|socket" | head -8
for rep in a b c d; do
for z in 0 1; do
if [ $z = 0 ]; then export SNF_BENCH_NO_NUMA=1; else unset SNF_BENCH_NO_NUMA; fi
timeout 300 python bench.py $Q > $O/c1_$z$rep.json 2> $O/c1_$z$rep.err
python - <<PY
import json
d=json.load(open('$O/c1_$z$rep.json')); print('numa_bind $z', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['config'].get('host_binding'))
PY
done
done
