#!/bin/bash
# round 5, first GPU session: the GPU suite on the new front end (w4s_segment / w6t_emit) and d1g_refine<8>, a same-box A/B against
# round 4's library (variants/base_r04.so), the launch timeline
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: nod1g:SNF_NO_D1_GROUPS=1 div2:SNF_GRID_DIV=2 2>&1 | tee gpurun_out/ab_r05_1.log
bash tools/timeline1.sh > /dev/null 2>&1; head -60 gpurun_out/timeline1.txt
