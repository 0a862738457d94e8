#!/bin/bash
# consensus-kernel iteration on one box: parity of the consensus entry point, per-workgroup trace (alone on the device), two bench lines
# usage: tools/r03_cons_ab.sh <tag> [trace variant name]
tag=$1; tv=${2:-trace}
O=gpurun_out/cons_ab; mkdir -p $O
python -m pytest tests/test_consensus_api.py -q -m gpu 2>&1 | tail -1 > $O/$tag.tests.txt
SNF_LIB_SO=$PWD/variants/$tv.so WG_SERIAL=1 bash tools/wg_trace.sh > $O/$tag.trace.txt
python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5 > $O/$tag.bench2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5 --inflight 1 > $O/$tag.bench1.json 2>/dev/null
