O=gpurun_out/dist1; mkdir -p $O
B="python bench.py --gpus 1 --steps 30 --warmup 5 --no-configs --no-cpu-baseline --no-wall-clock"
$B > $O/m_own.json 2>/dev/null
SNF_BENCH_RESMEM=numpy $B > $O/m_numpy.json 2> $O/m_numpy.err
SNF_BENCH_RESMEM=shm $B > $O/m_shm.json 2> $O/m_shm.err
$B > $O/m_own2.json 2>/dev/null
