O=gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for rep in 1 2; do
$B > $O/sal0_2_$rep.json 2>/dev/null
SNF_SMALL_AFTER_LARGE=1 $B > $O/sal1_2_$rep.json 2>/dev/null
$B --inflight 1 > $O/sal0_1_$rep.json 2>/dev/null
SNF_SMALL_AFTER_LARGE=1 $B --inflight 1 > $O/sal1_1_$rep.json 2>/dev/null
done
SNF_SMALL_AFTER_LARGE=1 SNF_ALT_HBM=1 $B > $O/sal1hbm_2_1.json 2>/dev/null
SNF_SMALL_AFTER_LARGE=1 SNF_ALT_HBM=1 $B --inflight 1 > $O/sal1hbm_1_1.json 2>/dev/null
