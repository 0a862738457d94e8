#!/bin/bash
# per-workgroup start / duration of the consensus kernels: needs a library built with -DSNF_WG_TRACE (SNF_LIB_SO=<that build>)
O=gpurun_out/wg_trace; mkdir -p $O
export TMPDIR=/tmp
export SNF_PROF=1; [ -n "$WG_SERIAL" ] && export SNF_SERIAL=1; timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 2 --warmup 1 --inflight 1 > $O/run${WG_SERIAL:+_serial}.json 2> $O/run${WG_SERIAL:+_serial}.err
grep SNF_WG_TRACE $O/run${WG_SERIAL:+_serial}.err | tail -44
