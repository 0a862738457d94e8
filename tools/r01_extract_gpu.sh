#!/bin/bash
# One GPU call: extraction parity on the MI355X, its measurement and kernel trace, then the default bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x1
rm -rf $O; mkdir -p $O
cd $R
S=$O/status.txt
date +%s > $S
timeout 300 python -m pytest tests/test_extract_gpu.py -x -q -m gpu > $O/pytest_extract.log 2>&1; echo "pytest_extract rc=$? t=$(date +%s)" >> $S
tail -3 $O/pytest_extract.log
timeout 150 python tools/bench_extract.py > $O/extract_bench.json 2> $O/extract_bench.err; echo "bench_extract rc=$? t=$(date +%s)" >> $S
cat $O/extract_bench.json
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x1 -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > $O/prof.log 2>&1; echo "rocprof rc=$? t=$(date +%s)" >> $S
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -12
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -size +2M -delete
timeout 240 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? t=$(date +%s)" >> $S
cat $O/bench.json
timeout 400 python -m pytest tests -x -q -m gpu --deselect tests/test_extract_gpu.py > $O/pytest_all.log 2>&1; echo "pytest_all rc=$? t=$(date +%s)" >> $S
tail -3 $O/pytest_all.log
cat $S
