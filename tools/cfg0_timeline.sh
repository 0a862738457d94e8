cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/cfg0; rm -rf $O; mkdir -p $O
for fl in 1 2; do python bench.py --config 0 --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 60 --warmup 10 --inflight $fl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config0 inflight $fl ms_per_step', d['ms_per_step'])"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o k -- python bench.py --config 0 --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 12 --warmup 4 > $O/run.log 2>&1
python tools/timeline.py $(find $O/t -name '*kernel_trace.csv' | head -1) > $O/timeline.txt 2>&1; cat $O/timeline.txt; rm -rf $O/t
