O=gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q $B > $O/q${q}_2.json 2>/dev/null
GPU_MAX_HW_QUEUES=$q $B --inflight 3 > $O/q${q}_3.json 2>/dev/null
GPU_MAX_HW_QUEUES=$q $B --inflight 1 > $O/q${q}_1.json 2>/dev/null
done
$B > $O/qdef_2.json 2>/dev/null
