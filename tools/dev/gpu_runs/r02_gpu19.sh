#!/bin/bash
O=gpurun_out/r02s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "consensus or golden or oracle or parity" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/serial.json 2> $O/serial.err
timeout 300 python bench.py $Q > $O/c1.json 2> $O/c1.err
for c in serial c1; do python -c "
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d['roofline']['gpu_ms_all_kernels'], [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:8]])"; done
