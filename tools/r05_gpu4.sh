#!/bin/bash
# round 5, fourth GPU session: the staged result with the copies at the fetch (default) and as copy kernels inside the pass, the consensus
# order under it, the per-task split with the library's own upload split, worker processes with a bounded number of hardware queues
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_output_modes.py tests/test_dropin_api.py tests/test_abi.py -m gpu -x -q > gpurun_out/pytest_gpu4.log 2>&1; tail -3 gpurun_out/pytest_gpu4.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: kcopy:SNF_STAGE_COPY=kernel order0:SNF_CONS_ORDER=0 kcopy0:SNF_STAGE_COPY=kernel,SNF_CONS_ORDER=0 2>&1 | tee gpurun_out/ab_r05_4.log
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -8
timeout 600 python tools/bench_workers.py 4 8 24 q0 q2 2>&1 | grep -v "^\[" | cut -c1-260 | tee gpurun_out/workers_4.log
