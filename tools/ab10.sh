mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline $cfg 2>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), round(d['value']/1e6,1), d['config']['signatures'], [(k['name'][-14:],round(k['ms'],3)) for k in d['roofline']['top_kernels'][:5]])" >> gpurun_out/ab.log 2>&1
  tail -1 gpurun_out/ab.err | cut -c1-200 >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
