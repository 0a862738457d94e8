#!/bin/bash
# tools/cons_ab.sh TAG[:ENV=VAL,...] ...: HIP-event times of the consensus kernels (bench.py's roofline.top_kernels) per variant,
# one batch in flight: in place, and with SNF_SERIAL=1 (every ALT kernel alone on the device)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/cons_ab; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 12 --warmup 3 --inflight 1"
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
  envs=$(echo $envs | tr ',' ' ')
  env $envs $B > $O/${tag}_place.json 2> $O/${tag}_place.err
  env $envs SNF_SERIAL=1 $B > $O/${tag}_serial.json 2> $O/${tag}_serial.err
  python - <<PY
import json
def row(f):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = {x["name"]: x["ms"] for x in d["roofline"]["top_kernels"]}
        return "step %.3f small %.3f large %.3f f4 %.3f" % (d["ms_per_step"], k.get("e45w_consensus_small", -1), k.get("e45w_consensus_large", -1), k.get("f4_emit", -1))
    except Exception as e: return "failed %r" % (e,)
print("$tag in place:", row("$O/${tag}_place.json"), "| alone:", row("$O/${tag}_serial.json"))
PY
done
