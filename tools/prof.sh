cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof/bench.log 2>&1
ls -R gpurun_out/prof | head -20
KT=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $KT > gpurun_out/prof/timeline.txt 2>&1
tail -5 gpurun_out/prof/timeline.txt
