# like ab.sh but prints the top kernels too
mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), [(k['name'][:14],round(k['ms'],3)) for k in d['roofline']['top_kernels'][:4]])" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
