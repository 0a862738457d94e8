#!/bin/bash
# tools/pmc.sh [bench flags]: HBM traffic per kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately, kernel-trace only)
#   -> gpurun_out/pmc_traffic.json (tools/pmc_parse.py);  e.g. bash tools/pmc.sh --genomes 4
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C; mkdir -p gpurun_out/pmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify "$@" > gpurun_out/pmc_$C.log 2>&1
done
python tools/pmc_parse.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
python - <<PY
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))["kernels"]
per = {k: v for k, v in d.items() if v["launches"] in (3, 4) and not k.startswith(("__amd", "rocprim"))}
lo = sum(v["hbm_bytes_lo"] for v in per.values()); hi = sum(v["hbm_bytes"] for v in per.values())
print("kernels of a pass: %d; HBM bytes per pass between %.0f and %.0f MB" % (len(per), lo / 1e6, hi / 1e6))
for k, v in sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:12]: print("  %-34s %7.1f .. %7.1f MB" % (k, v["hbm_bytes_lo"] / 1e6, v["hbm_bytes"] / 1e6))
PY
