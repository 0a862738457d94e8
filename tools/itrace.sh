#!/bin/bash
# per-workgroup start / duration of the main kernels of a pass: needs a library built with -DSNF_ITRACE
# (bash tools/build_variant.sh itrace -DSNF_ITRACE; SNF_LIB_SO=variants/itrace.so).  One batch in flight, then two.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/itrace; mkdir -p $O
export SNF_PROF=1 SNF_LIB_SO=$R/variants/itrace.so
for fl in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 3 --warmup 2 --inflight $fl > $O/run_$fl.json 2> $O/run_$fl.err
  grep SNF_ITRACE $O/run_$fl.err | tail -15 > $O/itrace_$fl.txt
  echo "== $fl in flight"; cat $O/itrace_$fl.txt
done
