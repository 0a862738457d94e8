#!/bin/bash
# same-box A/B: lead aggregates carried from d2w_call (this tree) against the previous commit's library (variants/prev.so)
O=gpurun_out/agg; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/${tag}_2.json 2>/dev/null; env "$@" $B --inflight 1 > $O/${tag}_1.json 2>/dev/null; }
run prev SNF_LIB_SO=$PWD/variants/prev.so
run new A=1
run new_d4 SNF_OCC_D2=4
run new_b8 SNF_E1_BATCH=8
run new_b32 SNF_E1_BATCH=32
run new_b64 SNF_E1_BATCH=64
run prev2 SNF_LIB_SO=$PWD/variants/prev.so
run new2 A=1
python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 10 --warmup 2 --no-verify > /dev/null 2>&1
