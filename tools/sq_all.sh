#!/bin/bash
# instruction counts, wave residency and LDS bank conflicts of EVERY kernel of a pass (one batch in flight): three rocprofv3 --pmc passes, kernel-trace only
# -> gpurun_out/sq_all/summary.txt: per kernel the duration, the time its VALU instructions alone need on 1024 SIMDs (4 cycles each),
#    the mean number of resident waves, instruction mix
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/sq_all; rm -rf $O; mkdir -p $O
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o sq -- python bench.py --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify "$@" > $O/p$i.log 2>&1
done
python tools/sq_all_parse.py $O/p1 $O/p2 $O/p3 > $O/summary.txt 2>&1; cat $O/summary.txt
rm -rf $O/p1 $O/p2 $O/p3
