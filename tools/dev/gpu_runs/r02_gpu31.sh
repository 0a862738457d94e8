#!/bin/bash
O=gpurun_out/r02aq; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c1.json 2> $O/c1.err
tail -2 $O/c1.err
python - <<PY
import json
d=json.load(open('$O/c1.json')); print(round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified')); print(d['wall_clock'])
PY
