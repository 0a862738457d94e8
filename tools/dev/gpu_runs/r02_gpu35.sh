#!/bin/bash
O=gpurun_out/r02aw; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/deepfuzz.py 40 > $O/deepfuzz.log 2>&1; tail -5 $O/deepfuzz.log
timeout 600 python tools/dev/bigfuzz.py 600 > $O/bigfuzz.log 2>&1; tail -3 $O/bigfuzz.log
