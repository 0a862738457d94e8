#!/bin/bash
# consensus kernels with phases switched off (SNF_DBG_ABLATE bits: 1 reads, 2 table build, 4 vote, 8 everything after the
# descriptor, 16 staging of the best read, 32 zeroing the counters); isolated launches.  Results are wrong by design.
O=gpurun_out/r02u; mkdir -p $O
export TMPDIR=/tmp SNF_BENCH_TOPK=40
Q="--no-cpu-baseline --no-wall-clock --steps 10 --warmup 3 --inflight 1"
for a in 0 8 7 23 55 3 1; do
  SNF_DBG_ABLATE=$a SNF_SERIAL=1 timeout 300 python bench.py $Q > $O/abl_$a.json 2> $O/abl_$a.err
  python - <<PY
import json
d=json.load(open('$O/abl_$a.json'))
print('ablate $a', [(k['name'],k['ms']) for k in d['roofline']['top_kernels'] if 'cons' in k['name']])
PY
done
