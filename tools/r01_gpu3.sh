#!/bin/bash
# GPU call 3: VCF / pipeline / SNF GPU tests, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x3
rm -rf $O; mkdir -p $O
cd $R
S=$O/status.txt
date +%s > $S
timeout 240 python -m pytest tests/test_pipeline.py tests/test_vcf.py tests/test_snf.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest_new rc=$? t=$(date +%s)" >> $S
tail -5 $O/pytest_new.log
timeout 300 python -m pytest tests -x -q -m gpu --deselect tests/test_pipeline.py --deselect tests/test_vcf.py --deselect tests/test_snf.py > $O/pytest_rest.log 2>&1; echo "pytest_rest rc=$? t=$(date +%s)" >> $S
tail -5 $O/pytest_rest.log
cat $S
