#!/bin/bash
set -x
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py --config 4 > $O/bench_c4.json 2> $O/bench_c4.err
tail -3 $O/bench_c4.err
python -c "
import json
d=json.load(open('$O/bench_c4.json')); print(round(d['value']), d['ms_per_step'], d.get('verified')); print(d['config'].get('host_phases_ms')); print(d['config']['rank0']); print(d['cpu_baseline']['value'])"
