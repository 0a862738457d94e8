# HBM traffic of the extraction kernels: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcx_$C; mkdir -p gpurun_out/pmcx_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcx_$C -o pmc -- python tools/bench_extract.py --steps 3 --cpu-reads 3 > gpurun_out/pmcx_$C/bench.log 2>&1
done
python tools/pmc_parse.py gpurun_out/pmcx_FETCH_SIZE gpurun_out/pmcx_WRITE_SIZE > gpurun_out/pmc_extract_traffic.json
python -c "
import json;d=json.load(open('gpurun_out/pmc_extract_traffic.json'))
for k,v in d['kernels'].items(): print(k, v)
"
find gpurun_out/pmcx_* -name '*.csv' -size +1M -delete
