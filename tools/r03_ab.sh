#!/bin/bash
# same-box A/B: the round-2 tree (.old_r02, commit a5f41a9, built in the build container) against the current tree, alternating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_ab; rm -rf $O; mkdir -p $O
one() { tag=$1; dir=$2; shift 2; (cd $dir && python bench.py --no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5 "$@" > $O/$tag.json 2> $O/$tag.err); python -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],3), d['config'].get('ms_per_pass_one_batch_in_flight'))"; }
for i in 1 2 3; do
  one old_$i $R/.old_r02
  one new_$i $R --no-configs
done
one old_c0 $R/.old_r02 --config 0
one new_c0 $R --no-configs --config 0
one old_c2 $R/.old_r02 --config 2
one new_c2 $R --no-configs --config 2
one old_c3 $R/.old_r02 --config 3
one new_c3 $R --no-configs --config 3
one new_cand $R --no-configs --output candidates
