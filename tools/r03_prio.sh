O=gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for rep in 1 2; do
$B > $O/prio_2_$rep.json 2>/dev/null
SNF_NO_STREAM_PRIO=1 $B > $O/noprio_2_$rep.json 2>/dev/null
$B --inflight 1 > $O/prio_1_$rep.json 2>/dev/null
SNF_NO_STREAM_PRIO=1 $B --inflight 1 > $O/noprio_1_$rep.json 2>/dev/null
done
SNF_ALT_HBM=1 $B > $O/priohbm_2_1.json 2>/dev/null
SNF_CONS_LARGE_NW=8 $B > $O/prionw8_2_1.json 2>/dev/null
