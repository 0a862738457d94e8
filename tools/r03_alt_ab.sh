O=gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for rep in 1 2; do
$B > $O/pin2_$rep.json 2>/dev/null; SNF_ALT_HBM=1 $B > $O/hbm2_$rep.json 2>/dev/null
$B --inflight 1 > $O/pin1_$rep.json 2>/dev/null; SNF_ALT_HBM=1 $B --inflight 1 > $O/hbm1_$rep.json 2>/dev/null
done
$B --inflight 3 > $O/pin3_1.json 2>/dev/null; SNF_ALT_HBM=1 $B --inflight 3 > $O/hbm3_1.json 2>/dev/null
