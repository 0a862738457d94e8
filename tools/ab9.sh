mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --inflight 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); ks={k['name']:k['ms'] for k in d['roofline']['top_kernels']}; print('$cfg', round(d['ms_per_step'],3), {k:round(v,3) for k,v in ks.items() if k in ('d2w_call','e1w_finalize','d1w_refine')})" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
