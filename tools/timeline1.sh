#!/bin/bash
# launch timeline of one pass of the default workload, one batch in flight -> gpurun_out/timeline1.txt   (extra bench flags: "$@")
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/tl1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o k -- python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 6 --warmup 2 "$@" > $O/run.log 2>&1
python tools/timeline.py $(find $O/t -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/timeline1.txt 2>&1; cat $R/gpurun_out/timeline1.txt; rm -rf $O
