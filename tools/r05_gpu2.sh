#!/bin/bash
# round 5, second GPU session: w4s_segment with tagged keys, the result staged through HBM (SNF_STAGE_OUT=1), f4w_emit on a small grid,
# SQ counters of the new kernels, the per-task split
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_prefilter.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu2.log 2>&1; tail -3 gpurun_out/pytest_gpu2.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: stage:SNF_STAGE_OUT=1 f4:SNF_F4_GRID=256 nod1g:SNF_NO_D1_GROUPS=1 2>&1 | tee gpurun_out/ab_r05_2.log
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for k in 1 2; do
  SNF_STAGE_OUT=1 $B --inflight 3 2>/dev/null | python -c "import json,sys; print('stage, three in flight', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"
  $B --inflight 3 2>/dev/null | python -c "import json,sys; print('new, three in flight', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"
done 2>&1 | tee -a gpurun_out/ab_r05_2.log
bash tools/sq_all.sh > /dev/null 2>&1; head -40 gpurun_out/sq_all/summary.txt
SNF_STAGE_OUT=1 bash tools/timeline1.sh > /dev/null 2>&1; cp gpurun_out/timeline1.txt gpurun_out/timeline1_stage.txt; head -50 gpurun_out/timeline1_stage.txt
timeout 300 python tools/per_task_prof.py 2>&1 | tail -4
