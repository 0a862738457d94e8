#!/bin/bash
# same-box A/B: lead aggregates carried from d2w_call (this tree) against the library before that change (variants/prev.so)
O=gpurun_out/agg; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/${tag}_2.json 2>/dev/null; env "$@" $B --inflight 1 > $O/${tag}_1.json 2>/dev/null; }
run prev SNF_LIB_SO=$PWD/variants/prev.so
run new A=1
run prev2 SNF_LIB_SO=$PWD/variants/prev.so
run new2 A=1
