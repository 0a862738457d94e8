#!/bin/bash
# round 5, ninth GPU session: how often do two passes in flight fall into step?  Ten runs of the headline with the pacing rules, ten without
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8 9 10; do
  for p in 1 0; do
    SNF_PACE=$p $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pace $p run $i: ms_per_step %.3f  LARGE %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done 2>&1 | tee gpurun_out/pace_r05.log
