#!/bin/bash
# hardware counters of the consensus kernels (serialised launches): one rocprofv3 --pmc pass per counter group, --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/cons_pmc; rm -rf $O; mkdir -p $O
B="python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --steps 2 --warmup 1"
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  SNF_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/g$i -o pmc -- $B > $O/g$i.log 2>&1
  f=$(find $O/g$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' >> $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "e45w_consensus" not in k and "e1w_finalize" not in k: continue
    k = "LARGE" if "ILi2E" in k else "SMALL" if "ILi1E" in k else k[:20]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    for c, v in sorted(d.items()): print(k, c, "%.4g" % v)
PY
done
cat $O/summary.txt
