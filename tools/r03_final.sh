#!/bin/bash
# round 3, final GPU run: the -m gpu suite, the default bench line exactly as the driver runs it, the profile set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r03_final; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputests.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['ms_per_step'], d['value'], d.get('verified'), d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['result_path']); print({k:(v.get('ms_per_step'),v.get('verified'),v.get('seconds')) for k,v in d.get('configs',{}).items()}); print(d['wall_clock']); print(d['cpu_baseline']['value'])"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --output candidates --no-configs > $O/bench_candidates.json 2> $O/bench_candidates.err
SNF_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-wall-clock > $O/bench_shared_1rank.json 2> $O/bench_shared_1rank.err; echo "shared rc=$?"
SNF_BENCH_FORCE_DIST=1 SNF_BENCH_GATHER=rccl timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-wall-clock > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; echo "rccl rc=$?"
SNF_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --scaling strong --no-configs --no-cpu-baseline --no-wall-clock > $O/bench_strong_1rank.json 2> $O/bench_strong_1rank.err; echo "strong rc=$?"
bash tools/r03_profile.sh final > $O/profile.log 2>&1
tools/probe/pcie_probe > $O/pcie.json 2>&1
