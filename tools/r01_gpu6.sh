#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x6
rm -rf $O; mkdir -p $O
cd $R
SNF_CONS_NW=1 timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_dropin_api.py tests/test_reference_vectors.py -x -q -m gpu > $O/pytest_nw1.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_nw1.log
for nw in 4 1; do
  SNF_CONS_NW=$nw timeout 100 python bench.py --no-cpu-baseline > $O/bench_nw$nw.json 2> $O/bench_nw$nw.err
  SNF_CONS_NW=$nw timeout 100 python bench.py --no-cpu-baseline --inflight 1 > $O/bench1_nw$nw.json 2>> $O/bench_nw$nw.err
  python - <<PY
import json
for f in ("$O/bench_nw$nw.json","$O/bench1_nw$nw.json"):
    d=json.load(open(f)); print("nw=$nw", f.split('/')[-1], d["ms_per_step"], [(k["name"],k["ms"],k.get("ms_one_batch_in_flight")) for k in d["roofline"]["top_kernels"][:3]])
PY
done
