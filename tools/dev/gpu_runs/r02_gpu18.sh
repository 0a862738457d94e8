#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
for nw in 4 8 16; do
  SNF_CONS_LARGE_NW=$nw SNF_SERIAL=1 timeout 300 python bench.py $Q --inflight 1 > $O/serial_$nw.json 2> $O/serial_$nw.err
  SNF_CONS_LARGE_NW=$nw timeout 300 python bench.py $Q > $O/c1_$nw.json 2> $O/c1_$nw.err
  SNF_CONS_LARGE_NW=$nw timeout 300 python bench.py --config 2 --no-cpu-baseline --no-wall-clock --steps 10 --warmup 3 > $O/c2_$nw.json 2> $O/c2_$nw.err
  for c in serial_$nw c1_$nw c2_$nw; do python -c "
import json
d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,1), round(d['ms_per_step'],3), d.get('verified'), [(k['name'],k['ms']) for k in d['roofline']['top_kernels'][:6]])"; done
done
