#!/bin/bash
# round 3, GPU call 1: where does the pass go?  Baseline lines, the LARGE-consensus spread experiment (ALT stores to HBM /
# no concurrent D2H / isolated launches), PCIe probe, untraced timelines, the driver command under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_exp1; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "
import json,sys
d=json.load(open('$O/$tag.json'))
tk={k['name']:(k['ms'],k.get('ms_one_batch_in_flight')) for k in d['roofline']['top_kernels']}
print(round(d['ms_per_step'],3), d['config'].get('ms_per_pass_one_batch_in_flight'), tk)
" 2>&1 | tail -1)"; }
tools/probe/pcie_probe > $O/pcie.json 2>&1; cat $O/pcie.json
export SNF_BENCH_TOPK=14
run base A=1
run alt_hbm SNF_ALT_HBM=1
run noprefetch SNF_PREFETCH=0
run alt_hbm_noprefetch SNF_ALT_HBM=1 SNF_PREFETCH=0
run serial SNF_SERIAL=1
run serial_alt_hbm SNF_SERIAL=1 SNF_ALT_HBM=1
run serial_alt_hbm_noprefetch SNF_SERIAL=1 SNF_ALT_HBM=1 SNF_PREFETCH=0
EXTRA="--inflight 1" run base_if1 A=1
EXTRA="--inflight 1" run alt_hbm_if1 SNF_ALT_HBM=1
EXTRA="--inflight 2" run base_if2 A=1
EXTRA="--config 0" run c0 A=1
EXTRA="--config 0 --inflight 1" run c0_if1 A=1
SNF_TIMELINE=1 python bench.py --no-cpu-baseline --no-wall-clock --inflight 1 --steps 3 --warmup 2 > $O/tl.json 2> $O/tl.err
SNF_TIMELINE=1 python bench.py --no-cpu-baseline --no-wall-clock --inflight 1 --steps 3 --warmup 2 --config 0 > $O/tl_c0.json 2> $O/tl_c0.err
# the driver command under rocprofv3 (kernel stats) and one batch in flight (trace -> timeline)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o k -- $B > $O/stats3.log 2>&1
cp $(find $O/stats3 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_3_in_flight.csv; grep '^{"metric"' $O/stats3.log | tail -1 > $O/bench_under_rocprof.json; rm -rf $O/stats3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o k -- python bench.py --no-cpu-baseline --no-wall-clock --inflight 1 --steps 6 --warmup 2 --config 0 > $O/stats1.log 2>&1
python tools/timeline.py $(find $O/stats1 -name '*kernel_trace.csv' | head -1) > $O/timeline_c0.txt 2>&1; rm -rf $O/stats1
grep -h SNF_TIMELINE $O/tl.err | tail -80 > $O/tl_last.txt
ls -la $O
