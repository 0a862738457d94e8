#!/bin/bash
# same-box A/B of the launch-shape knobs of the consensus kernels (two batches in flight, one in flight)
O=gpurun_out/knobs; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/${tag}_2.json 2>/dev/null; env "$@" $B --inflight 1 > $O/${tag}_1.json 2>/dev/null; }
run base A=1
run lnw8 SNF_CONS_LARGE_NW=8
run snw1 SNF_CONS_NW=1
run occ6 SNF_OCC_S=6
run occ8 SNF_OCC_S=8
run lnw8occ6 SNF_CONS_LARGE_NW=8 SNF_OCC_S=6
run base2 A=1
