#!/bin/bash
# final round-2 measurement set: GPU tests, the five bench configs as the driver would run them, strong scaling on one rank,
# then the profile set (kernel stats / timeline / PMC traffic / SQ counters).   bash tools/r02_final.sh <tag>
TAG=${1:-z}
O=gpurun_out/r02_$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config1.json 2> $O/bench_config1.err
for c in 0 2 3; do timeout 900 python bench.py --config $c > $O/bench_config$c.json 2> $O/bench_config$c.err; done
timeout 900 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err
SNF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scaling strong --no-cpu-baseline > $O/bench_strong_1rank_rccl.json 2> $O/bench_strong.err
for c in bench_config1 bench_config0 bench_config2 bench_config3 bench_config4 bench_strong_1rank_rccl; do python - <<PY
import json
try:
    d=json.load(open('$O/$c.json')); print('$c', round(d['value']/1e6,3), round(d['ms_per_step'],3), d.get('verified'), d['roofline']['kernel'], d['roofline']['frac'], (d.get('wall_clock') or {}).get('batched'))
except Exception as e: print('$c', 'FAILED', e)
PY
done
bash tools/r02_profile.sh r02$TAG > $O/profile.log 2>&1
