#!/bin/bash
# round 3 profile set of the default bench workload (BASELINE configs[1]):
#   kernel statistics of the EXACT command the driver runs (two batches in flight, the default) + the line that traced run printed,
#   kernel statistics + launch timeline with one batch in flight, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc
#   passes; --pmc is never combined with anything but --kernel-trace), and the hash of the kernel sources they belong to
#   (bench.py refuses to quote them for other sources).
# usage: bash tools/r03_profile.sh <tag>      -> gpurun_out/prof_<tag>/  (copy into profiles/ as r03_*)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
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
ls -la $O; head -c 400 $O/pmc_traffic.json; head -70 $O/timeline.txt
