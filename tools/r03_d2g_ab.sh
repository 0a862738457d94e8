#!/bin/bash
# same-box A/B: call_from by cluster size (d2g_call<8> / <32> / d2w_call) against a wave per cluster (SNF_NO_D2_GROUPS=1)
O=gpurun_out/d2g; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/${tag}_2.json 2>/dev/null; env "$@" $B --inflight 1 > $O/${tag}_1.json 2>/dev/null; }
run old SNF_NO_D2_GROUPS=1
run new A=1
run old2 SNF_NO_D2_GROUPS=1
run new2 A=1
python bench.py --no-wall-clock --no-configs --steps 10 --warmup 2 > $O/verify.json 2>/dev/null
python -m pytest tests/test_gpu_parity.py tests/test_output_modes.py -q -m gpu 2>&1 | tail -1
