#!/bin/bash
O=gpurun_out/r02aj; mkdir -p $O
export TMPDIR=/tmp
SNF_PROF=1 timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --steps 2 --warmup 1 --inflight 1 > $O/run.json 2> $O/run.err
grep "counts" $O/run.err | tail -1
