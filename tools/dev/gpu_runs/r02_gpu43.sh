#!/bin/bash
O=gpurun_out/r02be; mkdir -p $O
export TMPDIR=/tmp
for z in 0 1; do
if [ $z = 1 ]; then export SNF_NO_PREFETCH=1; else unset SNF_NO_PREFETCH; fi
timeout 900 python bench.py --config 4 --no-cpu-baseline --steps 4 --warmup 1 > $O/c4_$z.json 2> $O/c4_$z.err
python - <<PY
import json
d=json.load(open('$O/c4_$z.json')); print('no_prefetch $z', round(d['ms_per_step'],1), d['config'].get('host_phases_ms'))
PY
done
