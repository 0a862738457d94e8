# HBM traffic per kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately, kernel-trace only) -> gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C; mkdir -p gpurun_out/pmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$C/bench.log 2>&1
  ls gpurun_out/pmc_$C | head
done
python tools/pmc_parse.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
head -c 1500 gpurun_out/pmc_traffic.json
