#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x4
rm -rf $O; mkdir -p $O
cd $R
for f in 0 0.2 0.6; do
  timeout 100 python tools/bench_extract.py --sa-frac $f --cpu-reads 3 --steps 6 > $O/extract_sa_$f.json 2> $O/err_$f.txt
  python -c "import json;d=json.load(open('$O/extract_sa_$f.json'));print('$f',d['ms_count_pass'],d['ms_emit_pass'],d['algo_bytes'],d['signatures'])"
done
