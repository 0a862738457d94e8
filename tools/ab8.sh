mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  timeout 300 python bench.py --steps 90 --warmup 6 --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), round(d['value']/1e6,1), d['config'].get('ms_per_pass_one_batch_in_flight'))" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
