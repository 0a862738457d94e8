#!/bin/bash
# the N > 1 code with ONE rank on the GPU box: shared-memory landing (default) and the RCCL block gather
O=gpurun_out/dist1; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-wall-clock"
$B > $O/n1.json 2> $O/n1.err
SNF_BENCH_FORCE_DIST=1 $B > $O/shared.json 2> $O/shared.err; echo "shared rc=$?"
SNF_BENCH_FORCE_DIST=1 SNF_BENCH_GATHER=rccl $B > $O/rccl.json 2> $O/rccl.err; echo "rccl rc=$?"
python -m pytest tests/test_output_modes.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -2
