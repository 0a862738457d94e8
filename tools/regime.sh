#!/bin/bash
# tools/regime.sh N TAG[:ENV=VAL,...] ...: N default-shaped runs per variant; how often a run lands in the slow regime (two passes in step)
N=$1; shift
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
  envs=$(echo $envs | tr ',' ' ')
  vals=""
  for k in $(seq 1 $N); do
    v=$(env $envs python bench.py --no-cpu-baseline --no-wall-clock --no-configs --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); print(round(d['ms_per_step'],3))")
    vals="$vals $v"
  done
  python - <<PY
v = sorted(float(x) for x in "$vals".split())
slow = [x for x in v if x > 1.05]
print("$tag", "runs", len(v), "slow", len(slow), "median", v[len(v)//2], "min", v[0], "max", v[-1], "| slow:", slow)
PY
done
