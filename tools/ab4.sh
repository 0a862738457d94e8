mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline $cfg 2>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), round(d['value']/1e6,1), [(k['name'][-14:],round(k['ms'],3)) for k in d['roofline']['top_kernels'][:4]])" >> gpurun_out/ab.log
  tail -2 gpurun_out/ab.err >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
