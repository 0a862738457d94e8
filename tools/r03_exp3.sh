#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_exp3; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --gpus 1 --steps 20 --warmup 5"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "
import json,sys
d=json.load(open('$O/$tag.json'))
tk={k['name']:(k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels']}
print(round(d['ms_per_step'],3), d['config'].get('ms_per_pass_one_batch_in_flight'), d['roofline'].get('result_path',{}).get('bytes_per_pass'), tk)
" 2>&1 | tail -1)"; }
export SNF_BENCH_TOPK=14
run base A=1
run nospread SNF_PF_SPREAD=0
run nopf SNF_NO_PREFILTER=1
EXTRA="--output candidates" run cand A=1
EXTRA="--inflight 1" run if1 A=1
EXTRA="--inflight 1" run if1_nopf SNF_NO_PREFILTER=1
EXTRA="--inflight 2" run if2 A=1
EXTRA="--config 0" run c0 A=1
EXTRA="--config 0 --inflight 1" run c0_if1 A=1
EXTRA="--config 2" run c2 A=1
EXTRA="--config 3" run c3 A=1
run base2 A=1
SNF_TIMELINE=1 python bench.py --no-cpu-baseline --no-wall-clock --no-configs --inflight 1 --steps 3 --warmup 2 > $O/tl.json 2> $O/tl.err
grep -h SNF_TIMELINE $O/tl.err | tail -70 > $O/tl_last.txt
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gputests.log
bash tools/r03_profile.sh exp3 nopmc > $O/profile.log 2>&1
ls $O | head -5
