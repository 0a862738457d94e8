# A/B of env knobs on one box: usage: bash tools/ab.sh "VAR1=a VAR2=b" "VAR1=c" ...
mkdir -p gpurun_out; : > gpurun_out/ab.log
for rep in 1 2; do
for cfg in "$@"; do
  r=$(env $cfg timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$cfg $r" >> gpurun_out/ab.log
done; done
cat gpurun_out/ab.log
