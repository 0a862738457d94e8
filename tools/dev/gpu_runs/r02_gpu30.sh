#!/bin/bash
O=gpurun_out/r02ap; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/c1.json 2> $O/c1.err
grep "SNF_PROF" $O/c1.err | tail -6
