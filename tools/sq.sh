# SQ counters of the wave kernels (isolated consensus launches): three rocprofv3 --pmc passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/sq_$i; mkdir -p gpurun_out/sq_$i
  SNF_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/sq_$i -o sq -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sq_$i/bench.log 2>&1
done
python tools/sq_parse.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 | tee gpurun_out/sq_summary.txt
