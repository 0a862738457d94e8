#!/bin/bash
# launch timeline of one pass, one batch in flight (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --steps 6 --warmup 2 > $O/stats.log 2>&1
KT=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $KT > $O/timeline.txt 2>&1
rm -rf $O/stats
