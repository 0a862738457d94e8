#!/bin/bash
# round 5, tenth GPU session: which pacing rule keeps two passes out of step at what price (eight runs each)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8; do
  for v in "SNF_PACE=2" "SNF_PACE=3" "SNF_PACE=1 SNF_PACE_FRAC=0.25" "SNF_PACE=3 SNF_PACE_FRAC=0.25"; do
    env $v $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v run $i: ms_per_step %.3f' % d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/pace2_r05.log
