#!/bin/bash
# the scale knob: G genomes per batch (same workload replicated inside one batch), one and two batches in flight
O=gpurun_out/genomes; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 20 --warmup 3"
for g in 1 2 4; do
$B --genomes $g --inflight 1 > $O/g${g}_1.json 2>/dev/null
$B --genomes $g --inflight 2 > $O/g${g}_2.json 2>/dev/null
done
