#!/bin/bash
O=gpurun_out/r02ao; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or golden or dropin or abi" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for th in 32 16 64; do
SNF_PROF=1 SNF_UPLOAD_THREADS=$th timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/c1_$th.json 2> $O/c1_$th.err
grep "upload:" $O/c1_$th.err | tail -2
python - <<PY
import json
d=json.load(open('$O/c1_$th.json')); print('threads $th', round(d['ms_per_step'],3), d['wall_clock']['batched'])
PY
done
