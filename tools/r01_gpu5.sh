#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x5
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_extract_gpu.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
for f in 0.2 0.6; do
  timeout 100 python tools/bench_extract.py --sa-frac $f --cpu-reads 150 --steps 10 > $O/extract_sa_$f.json 2> $O/err_$f.txt
  python -c "import json;d=json.load(open('$O/extract_sa_$f.json'));print('$f',d['ms_count_pass'],d['ms_emit_pass'],d['roofline'],d['wall_ms_run_incl_scans_and_result_copy'])"
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x5 -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > $O/prof.log 2>&1
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -5 | cut -c1-160
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -size +2M -delete
