#!/bin/bash
# tools/profile.sh TAG [nopmc]: the profile set of the default bench workload (BASELINE configs[1]) -> gpurun_out/prof_TAG/
#   kernel_stats_default.csv   rocprofv3 --kernel-trace --stats of the EXACT command the driver runs (two batches in flight)
#   bench_under_rocprof.json   the line that traced run printed
#   kernel_stats.csv, timeline.txt   one batch in flight: kernel statistics and the launch timeline of one pass (tools/timeline.py)
#   pmc_traffic.json           HBM traffic: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never combined with anything but
#                              --kernel-trace), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/pmc_parse.py)
#   profile_meta.json          hash of the kernel sources the files belong to (bench.py refuses to quote them for other sources)
# Copy what is to be judged into profiles/ as rNN_*.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
python -c "from sniffles_amd import build; import json; print(json.dumps(dict(csrc_sha=build._lib_digest(), command='python bench.py --gpus 1 --steps 20 --warmup 5 (CPU legs skipped)')))" > $O/profile_meta.json
D="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --gpus 1 --steps 20 --warmup 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o k -- $D > $O/stats3.log 2>&1
cp $(find $O/stats3 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_default.csv; grep '^{"metric"' $O/stats3.log | tail -1 > $O/bench_under_rocprof.json; rm -rf $O/stats3
B="python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $B --steps 6 --warmup 2 > $O/stats.log 2>&1
KT=$(find $O/stats -name '*kernel_trace.csv' | head -1); ST=$(find $O/stats -name '*kernel_stats.csv' | head -1)
python tools/timeline.py $KT > $O/timeline.txt 2>&1; cp $ST $O/kernel_stats.csv
if [ "$2" != "nopmc" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o pmc -- $B --steps 2 --warmup 1 > $O/pmc_$C.log 2>&1
done
python tools/pmc_parse.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json
fi
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O; head -c 400 $O/pmc_traffic.json 2>/dev/null; head -70 $O/timeline.txt
