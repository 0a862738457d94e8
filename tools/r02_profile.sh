#!/bin/bash
# round 2 profile set of the default bench workload (BASELINE configs[1]), one batch in flight:
#   rocprofv3 kernel stats + timeline, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ counters incl.
#   SQ_LDS_BANK_CONFLICT (three --pmc passes, isolated launches).  --pmc is never combined with anything but --kernel-trace.
# usage: bash tools/r02_profile.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock"
# kernel statistics of the bench command itself (three batches in flight, as the driver runs it; only the CPU legs are skipped):
# the average durations here are what roofline.kernel_ms (mean over the timed passes) has to agree with
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o k -- python bench.py --no-cpu-baseline --no-wall-clock --gpus 1 --steps 20 --warmup 5 > $O/stats3.log 2>&1
cp $(find $O/stats3 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_3_in_flight.csv; grep '^{"metric"' $O/stats3.log | tail -1 > $O/bench_under_rocprof.json; rm -rf $O/stats3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $B --steps 6 --warmup 2 > $O/stats.log 2>&1
KT=$(find $O/stats -name '*kernel_trace.csv' | head -1); ST=$(find $O/stats -name '*kernel_stats.csv' | head -1)
python tools/timeline.py $KT > $O/timeline.txt 2>&1; cp $ST $O/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o pmc -- $B --steps 2 --warmup 1 > $O/pmc_$C.log 2>&1
done
python tools/pmc_parse.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  SNF_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sq_$i -o sq -- $B --steps 2 --warmup 1 > $O/sq_$i.log 2>&1
done
python tools/sq_parse.py $O/sq_1 $O/sq_2 $O/sq_3 > $O/sq_summary.txt 2>&1
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq_1 $O/sq_2 $O/sq_3     # raw traces are large; the summaries stay
ls -la $O; head -c 600 $O/pmc_traffic.json; head -30 $O/sq_summary.txt; head -40 $O/timeline.txt
