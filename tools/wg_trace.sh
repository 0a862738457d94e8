#!/bin/bash
# per-workgroup start / duration / hardware slot of the consensus kernels: needs a library built with -DSNF_WG_TRACE
# (tools/build_variant.sh trace -DSNF_WG_TRACE; SNF_LIB_SO=variants/trace.so).  Two runs: in place, and every ALT kernel alone
# (SNF_SERIAL=1); the raw per-workgroup table of the last pass goes to gpurun_out/wg_trace/trace_{place,serial}.txt
O=gpurun_out/wg_trace; mkdir -p $O
export TMPDIR=/tmp
export SNF_PROF=1
for mode in place serial; do
  [ $mode == serial ] && export SNF_SERIAL=1
  SNF_WG_TRACE_FILE=$O/trace_$mode.txt timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 2 --warmup 1 --inflight 1 > $O/run_$mode.json 2> $O/run_$mode.err
  grep SNF_WG_TRACE $O/run_$mode.err | tail -44 > $O/summary_$mode.txt
done
tail -44 $O/summary_serial.txt
