#!/bin/bash
# tools/run_ab.sh [-n ROUNDS] [-b "extra bench flags"] TAG[:ENV=VAL[,ENV=VAL...]] ...
# Same-box A/B of the default bench line: the variants run alternately ROUNDS times (default 2), two batches in flight and one,
# results under gpurun_out/ab/<tag>_<round>_{2,1}.json, one summary line per run.  A variant is a tag plus environment
# assignments - a knob of the library (SNF_NO_WINFRONT=1, SNF_WIN_BITS=9, SNF_NO_GRAPH=1 ...) or another build of the sources
# (SNF_LIB_SO=variants/x.so, see tools/build_variant.sh).
#   gpurun -- 'bash tools/run_ab.sh base: legacy:SNF_NO_WINFRONT=1'
ROUNDS=2; EXTRA=""
while getopts "n:b:" o; do case $o in n) ROUNDS=$OPTARG;; b) EXTRA=$OPTARG;; esac; done
shift $((OPTIND - 1))
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5 $EXTRA"
for i in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
    envs=$(echo $envs | tr ',' ' ')
    env $envs $B > $O/${tag}_${i}_2.json 2> $O/${tag}_${i}_2.err
    env $envs $B --inflight 1 > $O/${tag}_${i}_1.json 2> $O/${tag}_${i}_1.err
    python - <<PY
import json
def ms(f):
    try: return round(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"], 3)
    except Exception as e: return "failed"
print("$tag round $i: two in flight", ms("$O/${tag}_${i}_2.json"), " one in flight", ms("$O/${tag}_${i}_1.json"))
PY
  done
done
