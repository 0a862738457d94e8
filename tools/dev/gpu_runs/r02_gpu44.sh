#!/bin/bash
O=gpurun_out/r02bf; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "snf or pipeline or combine or dropin or genotype or vcf" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
