#!/bin/bash
# GPU call 2: new GPU tests (SNF, chain cuts, extraction after the NM-sum change), extraction and combine measurements
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x2
rm -rf $O; mkdir -p $O
cd $R
S=$O/status.txt
date +%s > $S
timeout 200 python -m pytest tests/test_snf.py tests/test_combine_task.py tests/test_extract_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest_new rc=$? t=$(date +%s)" >> $S
tail -5 $O/pytest_new.log
timeout 120 python tools/bench_extract.py > $O/extract_bench.json 2> $O/extract_bench.err; echo "bench_extract rc=$? t=$(date +%s)" >> $S
cat $O/extract_bench.json
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x2 -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > $O/prof.log 2>&1; echo "rocprof rc=$? t=$(date +%s)" >> $S
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -5 | cut -c1-200
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -size +2M -delete
timeout 200 python tools/bench_combine.py > $O/combine_bench.json 2> $O/combine_bench.err; echo "bench_combine rc=$? t=$(date +%s)" >> $S
cat $O/combine_bench.json; tail -3 $O/combine_bench.err
cat $S
