#!/bin/bash
O=gpurun_out/r02ay; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/combine_twin_gpu.py 0.5 10 > $O/twin.log 2>&1; tail -6 $O/twin.log
