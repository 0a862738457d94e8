#!/bin/bash
# round 5, eighth GPU session: knobs whose optimum may have moved with the staged result and the segment front end
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
bash tools/run_ab.sh -n 2 new: w9:SNF_WIN_BITS=9 w8:SNF_WIN_BITS=8 nw4:SNF_CONS_NW=4 nw4o0:SNF_CONS_NW=4,SNF_CONS_ORDER=0 mid:SNF_D2_MID=1 2>&1 | tee gpurun_out/ab_r05_5.log
