#!/bin/bash
# round 2, GPU call 2: LDS-vote consensus kernels - parity suite, bench, variants, timeline of the 60x HiFi config
set -x
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
SNF_CONS_NW=1 timeout 300 python bench.py $Q > $O/v_nw1.json 2> $O/v_nw1.err
SNF_OCC_S=6 timeout 300 python bench.py $Q > $O/v_occs6.json 2> $O/v_occs6.err
SNF_OCC_S=8 timeout 300 python bench.py $Q > $O/v_occs8.json 2> $O/v_occs8.err
SNF_OCC_D2=5 timeout 300 python bench.py $Q > $O/v_occd5.json 2> $O/v_occd5.err
timeout 300 python bench.py $Q --inflight 1 > $O/v_inflight1.json 2> $O/v_inflight1.err
SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/v_serial.json 2> $O/v_serial.err
SNF_TIMELINE=1 timeout 300 python bench.py --config 2 --no-cpu-baseline --no-wall-clock --steps 2 --warmup 1 --inflight 1 > $O/c2_timeline.json 2> $O/c2_timeline.err
SNF_PROF=1 timeout 300 python bench.py --config 2 --no-cpu-baseline --no-wall-clock --steps 2 --warmup 1 --inflight 1 > /dev/null 2> $O/c2_prof.err
for f in $O/v_*.json $O/bench_c1.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['ms_per_step'],3), d['config']['ms_per_pass_one_batch_in_flight'], [(k['name'],k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels'][:6]], d.get('verified'))"; done
grep -c TIMELINE $O/c2_timeline.err
