#!/bin/bash
# combine GPU tests after the table packing, combine measurement, default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x7
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_combine_task.py tests/test_combine.py tests/test_vcf.py tests/test_snf.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
SNF_BENCH_ORACLE_WINDOWS=20 timeout 150 python tools/bench_combine.py > $O/combine_bench.json 2> $O/combine_bench.err; echo "bench_combine rc=$?"
cat $O/combine_bench.json
