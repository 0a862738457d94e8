#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_exp4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputests.log
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --gpus 1 --steps 20 --warmup 5"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "
import json,sys
d=json.load(open('$O/$tag.json'))
tk={k['name']:(k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels']}
print(round(d['ms_per_step'],3), d['config'].get('ms_per_pass_one_batch_in_flight'), d['roofline'].get('result_path',{}).get('bytes_per_pass'), tk)
" 2>&1 | tail -1)"; }
export SNF_BENCH_TOPK=12
run base A=1
run nodefer SNF_NO_RN_DEFER=1
run serial SNF_SERIAL=1
EXTRA="--inflight 1" run if1 A=1
EXTRA="--config 0" run c0 A=1
run base2 A=1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2> $O/full.err; echo "full rc=$?"; python -c "
import json; d=json.load(open('$O/full.json')); print(d['ms_per_step'], d.get('verified'), d['roofline']['result_path'], {k:(v.get('ms_per_step'),v.get('verified'),v.get('seconds')) for k,v in d.get('configs',{}).items()}); print(d['wall_clock'])"
bash tools/r03_profile.sh exp4 > $O/profile.log 2>&1
