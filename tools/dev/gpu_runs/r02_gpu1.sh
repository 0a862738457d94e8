#!/bin/bash
# round 2, GPU call 1: parity suite, bench lines of configs 0-3, strong-scaling dry run, upload profile
set -x
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
SNF_PROF=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo rc=$? >> $O/bench_c1.err
cat $O/bench_c1.json | head -c 3000
grep -E "upload:|Traceback|Error" $O/bench_c1.err | head
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err
timeout 600 python bench.py --config 3 --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo rc=$? >> $O/bench_c3.err
timeout 300 python bench.py --config 0 --steps 20 --warmup 3 > $O/bench_c0.json 2> $O/bench_c0.err; echo rc=$? >> $O/bench_c0.err
timeout 300 python bench.py --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_strong.json 2> $O/bench_strong.err; echo rc=$? >> $O/bench_strong.err
SNF_BENCH_FORCE_DIST=1 timeout 300 python bench.py --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_strong_dist.json 2> $O/bench_strong_dist.err; echo rc=$? >> $O/bench_strong_dist.err
tail -3 $O/*.err
