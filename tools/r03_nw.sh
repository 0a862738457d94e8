O=gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
$B > $O/nw4_2.json 2>/dev/null
SNF_CONS_LARGE_NW=8 $B > $O/nw8_2.json 2>/dev/null
SNF_CONS_LARGE_NW=16 $B > $O/nw16_2.json 2>/dev/null
SNF_CONS_LARGE_NW=8 SNF_ALT_HBM=1 $B > $O/nw8hbm_2.json 2>/dev/null
SNF_CONS_LARGE_NW=8 $B --inflight 1 > $O/nw8_1.json 2>/dev/null
SNF_CONS_LARGE_NW=16 $B --inflight 1 > $O/nw16_1.json 2>/dev/null
SNF_CONS_GRID_MULT=1 $B > $O/gm1_2.json 2>/dev/null
$B > $O/nw4b_2.json 2>/dev/null
