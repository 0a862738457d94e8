#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_exp5; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --gpus 1 --steps 20 --warmup 5"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "
import json,sys
d=json.load(open('$O/$tag.json'))
print(round(d['ms_per_step'],3), d['config'].get('ms_per_pass_one_batch_in_flight'))
" 2>&1 | tail -1)"; }
for i in 1 2 3; do
EXTRA="--inflight 2" run if2_$i A=1
EXTRA="--inflight 3" run if3_$i A=1
done
EXTRA="--inflight 2 --steps 40" run if2_40 A=1
EXTRA="--inflight 3 --steps 40" run if3_40 A=1
EXTRA="--inflight 2 --config 0" run c0_if2 A=1
EXTRA="--inflight 3 --config 0" run c0_if3 A=1
EXTRA="--inflight 2 --config 2" run c2_if2 A=1
EXTRA="--inflight 3 --config 2" run c2_if3 A=1
EXTRA="--inflight 2 --config 3" run c3_if2 A=1
EXTRA="--inflight 3 --config 3" run c3_if3 A=1
