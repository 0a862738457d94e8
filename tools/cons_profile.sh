#!/bin/bash
# phase cycles of the consensus kernels: needs the library built with SNF_EXTRA_FLAGS=-DSNF_CONS_PROFILE (s_memtime stamps)
O=gpurun_out/cons_profile; mkdir -p $O
export TMPDIR=/tmp
SNF_PROF=1 SNF_SERIAL=1 timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --steps 2 --warmup 1 --inflight 1 > $O/run.json 2> $O/run.err
grep SNF_CONS_PROFILE $O/run.err | tail -14
