#!/bin/bash
# round 5, eleventh GPU session: list-fed wave kernels on larger grids (fewer clusters per wave, more waves in the dispatcher's hands)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
bash tools/run_ab.sh -n 2 new: gm2:SNF_GRID_MULT=2 gm4:SNF_GRID_MULT=4 2>&1 | tee gpurun_out/ab_r05_6.log
SNF_GRID_MULT=4 bash tools/timeline1.sh > /dev/null 2>&1; grep -E "d1g|d1w|d2g|d2w|e1w|span" gpurun_out/timeline1.txt
bash tools/timeline1.sh > /dev/null 2>&1; grep -E "d1g|d1w|d2g|d2w|e1w|span" gpurun_out/timeline1.txt
