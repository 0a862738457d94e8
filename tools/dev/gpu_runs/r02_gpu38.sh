#!/bin/bash
O=gpurun_out/r02az; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5"
for rep in a b; do
for z in 0 1; do
if [ $z = 1 ]; then export SNF_ALT_ZEROCOPY=1; else unset SNF_ALT_ZEROCOPY; fi
timeout 300 python bench.py $Q > $O/c1_$z$rep.json 2> $O/c1_$z$rep.err
python - <<PY
import json
d=json.load(open('$O/c1_$z$rep.json')); print('zero_copy $z', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:5]])
PY
done
done
