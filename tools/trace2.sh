#!/bin/bash
# kernel trace of the bench line with the default number of batches in flight -> tools/busy.py (how busy is the device, which kernels stretch)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/trace2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o k -- python bench.py --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 16 --warmup 4 "$@" > $O/run.log 2>&1
python tools/busy.py $(find $O/t -name '*kernel_trace.csv' | head -1) > $O/busy.txt 2>&1; cat $O/busy.txt
rm -rf $O/t
