#!/bin/bash
# tools/r06_sessions.sh N: the GPU sessions of round 6 (one `gpurun -- bash tools/r06_sessions.sh N` each), in the order they were run;
# what they wrote is under profiles/ (profiles/README.md, "Round 6").
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-wall-clock --no-configs"
ms() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$1', round(d['ms_per_step'],3), d.get('verified'), d.get('verified_vs_reference') if not isinstance(d.get('verified_vs_reference'), dict) else d['verified_vs_reference'].get('ok'))"; }
case "$1" in
1)
# round 6, first session: the population merge at its headline size with every merged record diffed against the unmodified reference's own
# writer (the reference leg now really runs the bit-parallel stand-in), the plain headline of this box as the round's starting point
( time timeout 1500 python bench.py --config 4 > gpurun_out/c4_full.json 2> gpurun_out/c4_full.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/c4_full.json").read().splitlines() if l.startswith('{"metric"')][-1])
print("config 4 ms_per_step", d["ms_per_step"], "verified", d.get("verified"), "kernel_ms", d["config"]["rank0"]["kernel_ms"])
print("verified_vs_reference", json.dumps(d.get("verified_vs_reference"))[:900])
cb = d.get("cpu_baseline", {})
print("reference", {k: cb.get(k) for k in ("kind", "hot_all_core_s", "hot_single_core_s", "cores", "vs_baseline", "same_population", "reference_error")})
PY
tail -3 gpurun_out/c4_full.err
python bench.py $Q --steps 40 --warmup 5 2>/dev/null | ms "headline two in flight"
python bench.py $Q --steps 40 --warmup 5 --inflight 1 2>/dev/null | ms "headline one in flight"
  ;;
2)
# round 6, second session: d4s_coverage (a thread per (call, sample), 16-ary rank descent) - the GPU parity tests, a same-box A/B against
# the former thread-per-call kernel (SNF_D4=thread), the launch timeline of one pass
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_output_modes.py tests/test_zz_gpu_end_to_end.py -m gpu -x -q > gpurun_out/pytest_gpu_2.log 2>&1; tail -3 gpurun_out/pytest_gpu_2.log
bash tools/run_ab.sh -n 2 old:SNF_D4=thread new: 2>&1 | tee gpurun_out/ab_r06_1.log
python bench.py $Q --genomes 4 --steps 20 --warmup 3 2>/dev/null | ms "genomes 4 new"
SNF_D4=thread python bench.py $Q --genomes 4 --steps 20 --warmup 3 2>/dev/null | ms "genomes 4 old"
bash tools/timeline1.sh > /dev/null 2>&1; head -70 gpurun_out/timeline1.txt
  ;;
3)
# round 6, third session: the ACGT column step of the wave Myers (Peq selects, two-bit carries through a DPP wave rotate, 32-bit
# bookkeeping): the DPP probe, the GPU tests of the merge path, config 4 against the library built before the change
hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_wave tools/probe/dpp_wave.hip 2>/dev/null && /tmp/dpp_wave
timeout 900 python -m pytest tests/test_edit_distance.py tests/test_combine.py tests/test_combine_task.py tests/test_pipeline.py -m gpu -x -q > gpurun_out/pytest_gpu_3.log 2>&1; tail -3 gpurun_out/pytest_gpu_3.log
for k in 1 2; do
  for tag in base new; do
    if [ $tag == base ]; then export SNF_LIB_SO=$R/variants/base_myers.so; else unset SNF_LIB_SO; fi
    python bench.py --config 4 --no-reference-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1])
print('$tag', 'ms_per_step', round(d['ms_per_step'],1), 'kernel_ms', d['config']['rank0']['kernel_ms'], 'abi', d['config']['rank0']['c_abi_call_ms'], 'verified', d.get('verified'), 'cells/s %.3g' % d['config']['rank0']['dp_cells_per_s'])"
  done
done 2>&1 | tee gpurun_out/ab_r06_2.log
unset SNF_LIB_SO
  ;;
4)
# round 6, fourth session: record-time columns + lazy calls at the seam: the GPU tests of the drop-in boundary, the default line without the
# side configs (wall_clock legs, worker processes, the reference beside them), P = 24 worker processes with a bounded number of GPU slots
timeout 900 python -m pytest tests/test_dropin_api.py tests/test_insitu_seam.py tests/test_output_modes.py tests/test_reference_pool.py tests/test_genotype.py tests/test_snf.py -m gpu -x -q > gpurun_out/pytest_gpu_4.log 2>&1; tail -3 gpurun_out/pytest_gpu_4.log
( time timeout 900 python bench.py --no-configs > gpurun_out/bench_noconf_4.json 2> gpurun_out/bench_noconf_4.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_noconf_4.json").read().splitlines() if l.startswith('{"metric"')][-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"))
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("batched", json.dumps(wc.get("batched")))
print("ingest", json.dumps(wc.get("ingest")))
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
for k, v in (wc.get("worker_processes") or {}).items():
    print("workers", k, v if not isinstance(v, dict) else {x: v[x] for x in ("hot_all_ms", "ingest_all_ms", "hw_queues_per_process", "n_out")})
PY
tail -3 gpurun_out/bench_noconf_4.err
for slots in 0 2 4 8; do timeout 240 python tools/workers_slots.py 24 $slots columns; done 2>&1 | grep '^{' | tee gpurun_out/workers_slots_4.log
  ;;
5)
# round 6, fifth session (bounded: every leg under its own timeout): the seam with the stand-in loops in C and no record copies; P = 24
# workers with GPU slots; LARGE consensus with the step words requested at the top of a read (A/B against the library before it); the N > 1
# code with one rank on the mixed gloo / RCCL group against the RCCL-only group of rounds 2-5
timeout 300 python -m pytest tests/test_dropin_api.py tests/test_insitu_seam.py -m gpu -x -q 2>&1 | tail -2
timeout 400 python bench.py --no-configs --no-reference-baseline --no-verify > gpurun_out/bench_wc_5.json 2> /dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_wc_5.json").read().splitlines() if l.startswith('{"metric"')][-1])
wc = d.get("wall_clock", {})
print("ms_per_step", d["ms_per_step"]); print("ingest", json.dumps(wc.get("ingest"))[:400])
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
for k, v in (wc.get("worker_processes") or {}).items():
    print("workers", k, v if not isinstance(v, dict) else {x: v[x] for x in ("hot_all_ms", "ingest_all_ms", "n_out")})
PY
for slots in 2 4 8; do timeout 200 python tools/workers_slots.py 24 $slots columns; done 2>&1 | grep '^{' | tee gpurun_out/workers_slots_5.log
timeout 150 python tools/workers_slots.py 24 4 leads 2>&1 | grep '^{' | tee -a gpurun_out/workers_slots_5.log
timeout 400 bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r06a.so new: 2>&1 | tee gpurun_out/ab_r06_3.log
for k in 1 2; do
  SNF_BENCH_FORCE_DIST=1 SNF_BENCH_PG=nccl timeout 120 python bench.py --gpus 1 $Q --steps 40 --warmup 5 2>/dev/null | ms "shared landing, one rank, RCCL-only group"
  SNF_BENCH_FORCE_DIST=1 timeout 120 python bench.py --gpus 1 $Q --steps 40 --warmup 5 2>/dev/null | ms "shared landing, one rank, gloo + lazy RCCL"
  timeout 120 python bench.py $Q --steps 40 --warmup 5 2>/dev/null | ms "plain line"
done 2>&1 | tee gpurun_out/shared_probe_5.log
  ;;
6)
# round 6, sixth session (every leg bounded): the GPU server for many workers, the eight-column Myers step against the one-column one
# (variants/myers1.so), the one-rank N > 1 line on both process groups again (three alternations)
timeout 600 python -m pytest tests/test_server.py tests/test_edit_distance.py tests/test_combine.py tests/test_combine_task.py tests/test_dropin_api.py -m gpu -x -q 2>&1 | tail -3
for spec in "24 server columns leads" "8 server columns" "4 server columns"; do timeout 300 python tools/workers_slots.py $spec; done 2>&1 | grep '^{' | tee gpurun_out/workers_server_6.log
for k in 1 2; do
  for tag in one eight; do
    if [ $tag == one ]; then export SNF_LIB_SO=$R/variants/myers1.so; else unset SNF_LIB_SO; fi
    timeout 300 python bench.py --config 4 --no-reference-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1])
print('$tag columns per step:', 'ms_per_step', round(d['ms_per_step'],1), 'kernel_ms', d['config']['rank0']['kernel_ms'], 'abi', d['config']['rank0']['c_abi_call_ms'], 'verified', d.get('verified'), 'cells/s %.3g' % d['config']['rank0']['dp_cells_per_s'])"
  done
done 2>&1 | tee gpurun_out/ab_r06_4.log
unset SNF_LIB_SO
for k in 1 2 3; do
  SNF_BENCH_FORCE_DIST=1 SNF_BENCH_PG=nccl timeout 120 python bench.py --gpus 1 $Q --steps 40 --warmup 5 2>/dev/null | ms "shared landing, one rank, RCCL-only group"
  SNF_BENCH_FORCE_DIST=1 timeout 120 python bench.py --gpus 1 $Q --steps 40 --warmup 5 2>/dev/null | ms "shared landing, one rank, gloo + lazy RCCL"
  timeout 120 python bench.py $Q --steps 40 --warmup 5 2>/dev/null | ms "plain line"
done 2>&1 | tee gpurun_out/shared_probe_6.log
  ;;
7)
# round 6, seventh session: the GPU server with two dispatchers and a gather window - its own split per batch (SNF_PROF), P = 24 / 8
SNF_PROF=1 timeout 300 python tools/workers_slots.py 24 server columns 2>&1 | grep -E '^\{|server batch' | cut -c1-250 | tee gpurun_out/workers_server_7.log
for spec in "24 server leads" "8 server columns" "24 server columns"; do timeout 300 python tools/workers_slots.py $spec; done 2>&1 | grep '^{' | tee -a gpurun_out/workers_server_7.log
  ;;
8)
# round 6, eighth session: the whole GPU suite on the final sources, then the profile set of the round (kernel statistics of the driver's
# command and of one batch in flight, launch timeline, HBM traffic at one and at four genomes per batch, SQ counters), extraction re-profiled
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_8.log 2>&1; tail -3 gpurun_out/pytest_gpu_8.log
timeout 1200 bash tools/profile.sh r06 > gpurun_out/profile_r06.log 2>&1; tail -5 gpurun_out/profile_r06.log
B="python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --genomes 4"
O=$R/gpurun_out/prof_r06
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc4_$C -o pmc -- $B --steps 2 --warmup 1 > $O/pmc4_$C.log 2>&1
done
python tools/pmc_parse.py $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE > $O/pmc_traffic_genomes4.json; rm -rf $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE
timeout 900 bash tools/sq_all.sh > /dev/null 2>&1; cp gpurun_out/sq_all/summary.txt $O/sq_all.txt; head -50 $O/sq_all.txt
timeout 300 python tools/bench_extract.py > $O/extract_bench.json 2> /dev/null; tail -c 1500 $O/extract_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/xstats -o k -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > $O/xstats.log 2>&1
cp $(find $O/xstats -name '*kernel_stats.csv' | head -1) $O/extract_kernel_stats.csv; rm -rf $O/xstats
timeout 600 bash tools/pmc_extract.sh > /dev/null 2>&1; cp gpurun_out/pmc_extract_traffic.json $O/extract_pmc_traffic.json
ls -la $O
  ;;
9)
# round 6, ninth session: the bench lines that are kept (tools/final_set.sh)
timeout 2000 bash tools/final_set.sh r06 2>&1 | tail -15
  ;;
10)
# round 6, tenth session: where the time of 24 object-fed workers goes through the GPU server (the slowest worker's split + the server's batches)
SNF_PROF=1 timeout 300 python tools/workers_slots.py 24 server leads 2>&1 | grep -E '^\{|server batch' | cut -c1-900 | tee gpurun_out/workers_server_10.log
timeout 300 python tools/workers_slots.py 24 server columns 2>&1 | grep '^{' | cut -c1-600 | tee -a gpurun_out/workers_server_10.log
  ;;
11)
# round 6, eleventh session: a lone pass with only its ALT bytes staged through HBM (SNF_ALT_HBM=1: the LARGE consensus workgroups no longer
# hold their LDS while ALT stores wait for PCIe; one copy at the fetch) against the direct stores, three alternations
for k in 1 2 3; do
  timeout 120 python bench.py $Q --steps 30 --warmup 5 --inflight 1 2>/dev/null | ms "one in flight, direct stores"
  SNF_ALT_HBM=1 timeout 120 python bench.py $Q --steps 30 --warmup 5 --inflight 1 2>/dev/null | ms "one in flight, ALT through HBM"
done 2>&1 | tee gpurun_out/ab_r06_5.log
SNF_ALT_HBM=1 timeout 120 python bench.py $Q --steps 30 --warmup 5 2>/dev/null | ms "two in flight with SNF_ALT_HBM (no effect expected: staged anyway)" | tee -a gpurun_out/ab_r06_5.log
  ;;
12)
# round 6, twelfth session: the default line once more on the final tree (what the driver runs), timed, and what it leaves in /dev/shm
( time timeout 1200 python bench.py > gpurun_out/bench_default_12.json 2> gpurun_out/bench_default_12.err ) 2>&1 | grep real
ls /dev/shm | head; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_default_12.json").read().splitlines() if l.startswith('{"metric"')][-1])
print("ms_per_step", d["ms_per_step"], d["config"]["verified"], d["config"]["verified_vs_reference"], d["config"]["configs_verified"])
print(json.dumps(d["cpu_baseline"]["vs_baseline"])[:700])
PY
  ;;
13)
# round 6, thirteenth session: the GPU server with its staging arena reserved up front (SNF_STAGE_ARENA_MB through server.start): 24 workers,
# three runs of each input form
for k in 1 2 3; do timeout 200 python tools/workers_slots.py 24 server columns leads; done 2>&1 | grep '^{' | cut -c1-330 | tee gpurun_out/workers_server_13.log
  ;;
14)
# round 6, fourteenth session: what an LDS double buffer would cost LARGE in occupancy - the kernel as it is with 24 KB of unused LDS per
# workgroup (one workgroup per CU instead of two; built from a copy of the sources, variants/large_1wg.so)
for k in 1 2; do
  for tag in base one_wg; do
    if [ $tag == one_wg ]; then export SNF_LIB_SO=$R/variants/large_1wg.so; else unset SNF_LIB_SO; fi
    for fl in 2 1; do
      timeout 120 python bench.py $Q --steps 40 --warmup 5 --inflight $fl 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1])
k=[x for x in d['roofline']['top_kernels'] if x['name']=='e45w_consensus_large'][0]
print('$tag in flight $fl: ms_per_step', round(d['ms_per_step'],3), 'LARGE ms', k['ms'])"
    done
  done
done 2>&1 | tee gpurun_out/ab_r06_6.log
unset SNF_LIB_SO
  ;;
15)
# round 6, fifteenth session: what the two-in-flight step is sensitive to - ablations of whole stages (not results: the lines say so)
for spec in "plain:" "symbolic:SNF_BENCH_CFG={\"symbolic\":true}" "no_consensus:SNF_BENCH_CFG={\"no_consensus\":true}" "three_in_flight:X=1" "pace_off:SNF_PACE=0" "plain2:"; do
  tag=${spec%%:*}; envs=${spec#*:}; fl=2; [ $tag == three_in_flight ] && fl=3
  for k in 1 2; do
    env $envs timeout 120 python bench.py $Q --no-verify --steps 40 --warmup 5 --inflight $fl 2>/dev/null | ms "$tag"
  done
done 2>&1 | tee gpurun_out/ab_r06_7.log
  ;;
16)
# round 6, sixteenth session: the extraction kernels rebuilt around the record's chain of dependent accesses (fixed fields as one load,
# clip operations + auxiliary region into LDS + two CIGAR steps in one round trip, tags as one 8-byte read each, the SA parser on the LDS
# copy): GPU tests of the extraction, then the same-box A/B against the library built from the sources before (variants/x_base.so)
# at four register budgets (SNF_EXTRACT_WAVES)
timeout 900 python -m pytest tests/test_extract_gpu.py tests/test_extract.py -m gpu -x -q > gpurun_out/pytest_gpu_16.log 2>&1; tail -3 gpurun_out/pytest_gpu_16.log
xb() { python tools/bench_extract.py --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3), 'frac', round(d['roofline']['count_pass']['frac'],4), round(d['roofline']['emit_pass']['frac'],4), 'leads', d['signatures'])"; }
for k in 1 2; do
  SNF_LIB_SO=$R/variants/x_base.so xb base
  for w in 4 5 6 8; do SNF_EXTRACT_WAVES=$w xb new_w$w; done
done 2>&1 | tee gpurun_out/ab_r06_8.log
SNF_EXTRACT_THREAD=1 xb thread_form | tee -a gpurun_out/ab_r06_8.log
  ;;
17)
# round 6, seventeenth session: what bounds the extraction passes - SQ counters of x_wave, and the same passes without SA tags, without
# any tag, on a table twice as long, on short reads
xb() { tag=$1; shift; python tools/bench_extract.py --steps 6 --cpu-reads 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'algo MB', round(d['algo_bytes']/1e6,1), 'records', d['records'], 'accepted', d['reads_accepted'], 'leads', d['signatures'])"; }
{
xb new_default
xb new_sa0 --sa-frac 0
xb new_notags --no-tags --sa-frac 0
xb new_tile16 --tile 16
xb new_short --read-len 3000 --reads 12000
SNF_LIB_SO=$R/variants/x_base.so xb base_sa0 --sa-frac 0
SNF_LIB_SO=$R/variants/x_base.so xb base_notags --no-tags --sa-frac 0
} 2>&1 | tee gpurun_out/ab_r06_9.log
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/sqx_$i; mkdir -p gpurun_out/sqx_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/sqx_$i -o sq -- python tools/bench_extract.py --steps 2 --cpu-reads 3 > gpurun_out/sqx_$i/bench.log 2>&1
done
python tools/sq_parse.py gpurun_out/sqx_1 gpurun_out/sqx_2 gpurun_out/sqx_3 | tee gpurun_out/sqx_summary.txt
find gpurun_out/sqx_* -name '*.csv' -size +1M -delete
  ;;
18)
# round 6, eighteenth session: the SA element in registers (it lived in scratch memory: the lane-0 parser waited a memory round trip per field)
xb() { tag=$1; shift; python tools/bench_extract.py --steps 6 --cpu-reads 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3), 'algo MB', round(d['algo_bytes']/1e6,1), 'records', d['records'], 'leads', d['signatures'])"; }
{
timeout 600 python -m pytest tests/test_extract_gpu.py -m gpu -x -q 2>&1 | tail -1
for k in 1 2; do
xb new_default
SNF_LIB_SO=$R/variants/x_base.so xb base_default
done
for w in 5 6 8; do SNF_EXTRACT_WAVES=$w xb new_w$w; done
xb new_sa0 --sa-frac 0
xb new_sa50 --sa-frac 0.5
xb new_tile16 --tile 16
xb new_tile32 --tile 32
} 2>&1 | tee gpurun_out/ab_r06_10.log
  ;;
19)
# round 6, nineteenth session: the SA string parsed by the wave (a lane per element) - GPU tests, per-record stamps, same-box A/B
xb() { tag=$1; shift; python tools/bench_extract.py --steps 6 --cpu-reads 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3), 'algo MB', round(d['algo_bytes']/1e6,1), 'records', d['records'], 'leads', d['signatures'])"; }
{
timeout 600 python -m pytest tests/test_extract_gpu.py tests/test_extract.py -m gpu -x -q 2>&1 | tail -1
SNF_LIB_SO=$R/variants/xtrace.so python tools/xtrace.py --pass count 2>&1 | grep -v "in flight at some"
SNF_LIB_SO=$R/variants/xtrace.so python tools/xtrace.py --pass emit 2>&1 | grep -v "in flight at some"
for k in 1 2; do
xb new_default
SNF_LIB_SO=$R/variants/x_base.so xb base_default
done
for w in 5 6; do SNF_EXTRACT_WAVES=$w xb new_w$w; done
xb new_sa0 --sa-frac 0
xb new_sa50 --sa-frac 0.5
xb new_tile32 --tile 32
SNF_LIB_SO=$R/variants/x_base.so xb base_tile32 --tile 32
} 2>&1 | tee gpurun_out/ab_r06_11.log
  ;;
20)
# round 6, twentieth session: is the extraction pass bound by the dispatch of 24 000 one-wave workgroups?  A capped grid that strides
xb() { tag=$1; shift; python tools/bench_extract.py --steps 6 --cpu-reads 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3), 'algo MB', round(d['algo_bytes']/1e6,1), 'records', d['records'], 'leads', d['signatures'])"; }
{
SNF_LIB_SO=$R/variants/xtrace.so python tools/xtrace.py --pass count 2>&1 | head -16
xb grid_all
for g in 2048 4096 8192 16384; do SNF_EXTRACT_GRID=$g xb grid_$g; done
xb grid_all_tile32 --tile 32
for g in 4096 8192 16384; do SNF_EXTRACT_GRID=$g xb grid_${g}_tile32 --tile 32; done
} 2>&1 | tee gpurun_out/ab_r06_12.log
  ;;
21)
# round 6, twenty-first session: the tag walk as scalar code, the CIGAR walk 2 / 3 / 4 steps ahead, the run without its forty allocations
xb() { tag=$1; shift; python tools/bench_extract.py --steps 8 --cpu-reads 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3), 'algo MB', round(d['algo_bytes']/1e6,1), 'records', d['records'], 'leads', d['signatures'])"; }
{
timeout 900 python -m pytest tests/test_extract_gpu.py tests/test_extract.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -1
for k in 1 2; do
SNF_LIB_SO=$R/variants/x_ahead2.so xb ahead2
xb ahead3
SNF_LIB_SO=$R/variants/x_ahead4.so xb ahead4
done
SNF_LIB_SO=$R/variants/x_ahead2.so xb ahead2_tile32 --tile 32
xb ahead3_tile32 --tile 32
SNF_LIB_SO=$R/variants/x_ahead4.so xb ahead4_tile32 --tile 32
SNF_LIB_SO=$R/variants/x_base.so xb base
} 2>&1 | tee gpurun_out/ab_r06_13.log
  ;;
22)
# round 6, twenty-second session: per-workgroup stamps of the main kernels (tools/itrace.sh) showed d1w_refine ending with a handful of
# waves on its largest clusters for half of its span; the hand-over lists now put items above SNF_HEAVY_N leads at the front.
# GPU parity tests, the stamps again, same-box A/B over the threshold (0 = one class, as before)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_end_to_end.py -m gpu -x -q 2>&1 | tail -1
bash tools/itrace.sh 2>&1 | grep -E "in flight$|d1w_refine|d2w_call|d2g_call|d1g_refine"
SNF_HEAVY_N=0 bash tools/itrace.sh 2>&1 | grep -E "in flight$|d1w_refine|d2w_call"
bash tools/run_ab.sh -n 2 h0:SNF_HEAVY_N=0 h24: h16:SNF_HEAVY_N=16 h32:SNF_HEAVY_N=32 2>&1 | tee gpurun_out/ab_r06_14.log
  ;;
23)
# round 6, twenty-third session: the call kernels are bound by instruction issue - sums that stay below 2^32 as 32-bit DPP reductions, the
# variance of a cluster whose values lie within 8191 of each other from two 32-bit sums and one fp64 division.  GPU parity, same-box A/B
# against the build before (variants/pre_sums.so), kernel times from the timeline
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_end_to_end.py tests/test_reference_vectors.py -m gpu -x -q 2>&1 | tail -1
bash tools/run_ab.sh -n 3 before:SNF_LIB_SO=$R/variants/pre_sums.so after: 2>&1 | tee gpurun_out/ab_r06_15.log
for so in $R/variants/pre_sums.so ""; do
  SNF_LIB_SO=$so SNF_TIMELINE=1 python bench.py $Q --no-verify --steps 3 --warmup 2 --inflight 1 2>&1 | grep -E "SNF_TIMELINE.*(d2g_call|d2w_call|d1w_refine)" | tail -3
done 2>&1 | tee -a gpurun_out/ab_r06_15.log
  ;;
24)
# round 6, twenty-fourth session: the streaming kernels of the front end with several items per thread (w1_hist / w3_scatter: four leads,
# w6t_emit: two positions) - their workgroups lived one or two round trips and the kernels were as many rounds of that as the device
# holds workgroups.  GPU parity, stamps, same-box A/B against the build before (variants/pre_wk.so), the timeline of the front end.
# RESULT: no gain (0.942-0.946 against 0.935-0.939 ms; the workgroups live three times as long, the kernels as long as before - they
# move ~26 B per lead at ~3 TB/s already) - taken back, the session stays as the record (profiles/ab_r06_16.log)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_prefilter.py tests/test_zz_gpu_end_to_end.py -m gpu -x -q 2>&1 | tail -1
bash tools/itrace.sh 2>&1 | grep -E "in flight$|w1_hist|w3_scatter|w6t_emit|w4s_segment" | head -5
bash tools/run_ab.sh -n 3 before:SNF_LIB_SO=$R/variants/pre_wk.so after: 2>&1 | tee gpurun_out/ab_r06_16.log
for so in $R/variants/pre_wk.so ""; do
  SNF_LIB_SO=$so SNF_TIMELINE=1 python bench.py $Q --no-verify --steps 3 --warmup 2 --inflight 1 2>&1 | grep -E "SNF_TIMELINE.*(w1_hist|w3_scatter|w6t_emit|w4s_segment)" | tail -4
done 2>&1 | tee -a gpurun_out/ab_r06_16.log
  ;;
25)
# round 6, twenty-fifth session: the whole GPU suite on the final sources, then the profile set of the round again (the kernel sources
# changed: extraction, hand-over lists, call kernels) - as session 8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_25.log 2>&1; tail -3 gpurun_out/pytest_gpu_25.log
timeout 1200 bash tools/profile.sh r06 > gpurun_out/profile_r06.log 2>&1; tail -5 gpurun_out/profile_r06.log
B="python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --genomes 4"
O=$R/gpurun_out/prof_r06
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc4_$C -o pmc -- $B --steps 2 --warmup 1 > $O/pmc4_$C.log 2>&1
done
python tools/pmc_parse.py $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE > $O/pmc_traffic_genomes4.json; rm -rf $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE
timeout 900 bash tools/sq_all.sh > /dev/null 2>&1; cp gpurun_out/sq_all/summary.txt $O/sq_all.txt; head -30 $O/sq_all.txt
timeout 300 python tools/bench_extract.py > $O/extract_bench.json 2> /dev/null; tail -c 1500 $O/extract_bench.json
timeout 300 python tools/bench_extract.py --tile 32 > $O/extract_bench_96000.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/xstats -o k -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > $O/xstats.log 2>&1
cp $(find $O/xstats -name '*kernel_stats.csv' | head -1) $O/extract_kernel_stats.csv; rm -rf $O/xstats
timeout 600 bash tools/pmc_extract.sh > /dev/null 2>&1; cp gpurun_out/pmc_extract_traffic.json $O/extract_pmc_traffic.json
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/sqx_$i; mkdir -p gpurun_out/sqx_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/sqx_$i -o sq -- python tools/bench_extract.py --steps 2 --cpu-reads 3 > gpurun_out/sqx_$i/bench.log 2>&1
done
python tools/sq_parse.py gpurun_out/sqx_1 gpurun_out/sqx_2 gpurun_out/sqx_3 > $O/extract_sq.txt; find gpurun_out/sqx_* -name '*.csv' -size +1M -delete
SNF_LIB_SO=$R/variants/xtrace.so python tools/xtrace.py --pass count > $O/xtrace_count.txt 2>&1
SNF_LIB_SO=$R/variants/xtrace.so python tools/xtrace.py --pass emit > $O/xtrace_emit.txt 2>&1
bash tools/itrace.sh > /dev/null 2>&1; cp gpurun_out/itrace/itrace_1.txt $O/itrace_1_in_flight.txt; cp gpurun_out/itrace/itrace_2.txt $O/itrace_2_in_flight.txt
ls -la $O
  ;;
26)
# round 6, twenty-sixth session: the bench lines that are kept (tools/final_set.sh), on the final sources
bash tools/final_set.sh r06 2>&1 | tail -14
  ;;
27)
# round 6, twenty-seventh session: fused-sequence copies of clusters with more than SNF_COPY_DEFER parts left to a copy kernel (the one
# workgroup d1w_refine ended with, 70 us, made sixteen rounds of four copies); the extraction's serial NM sum over a dense array
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_end_to_end.py tests/test_extract_gpu.py -m gpu -x -q 2>&1 | tail -1
bash tools/itrace.sh 2>&1 | grep -E "in flight$|d1w_refine"
SNF_COPY_DEFER=0 bash tools/itrace.sh 2>&1 | grep -E "in flight$|d1w_refine"
bash tools/run_ab.sh -n 3 inline:SNF_COPY_DEFER=0 defer8: defer16:SNF_COPY_DEFER=16 2>&1 | tee gpurun_out/ab_r06_17.log
for d in 0 8; do SNF_COPY_DEFER=$d SNF_TIMELINE=1 python bench.py $Q --no-verify --steps 3 --warmup 2 --inflight 1 2>&1 | grep -E "SNF_TIMELINE.*(d1w_refine|d1c_copy|d1g_refine)" | tail -3; done 2>&1 | tee -a gpurun_out/ab_r06_17.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xstats27 -o k -- python tools/bench_extract.py --steps 5 --cpu-reads 5 > gpurun_out/xstats27.log 2>&1
head -6 $(find gpurun_out/xstats27 -name '*kernel_stats.csv' | head -1) | cut -c1-120 | tee -a gpurun_out/ab_r06_17.log
python tools/bench_extract.py --steps 8 --cpu-reads 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extract count', round(d['ms_count_pass'],4), 'emit', round(d['ms_emit_pass'],4), 'run wall', round(d['wall_ms_run_incl_scans_and_result_copy'],3))" | tee -a gpurun_out/ab_r06_17.log
rm -rf gpurun_out/xstats27
  ;;
28)
# round 6, twenty-eighth session: LARGE consensus alone is as long as its longest call (itrace: the heaviest call runs 184 us of the kernel's
# 196) - eight waves per call (SNF_CONS_LARGE_NW=8) measured again under the staged result; then the default line once more, now that
# profiles/r06_profile_meta.json carries the hash of the built sources (the line quotes the rocprofv3 average only then)
bash tools/run_ab.sh -n 2 nw4: nw8:SNF_CONS_LARGE_NW=8 2>&1 | tee gpurun_out/ab_r06_18.log
python - <<'PY' | tee -a gpurun_out/ab_r06_18.log
import json, glob
for f in sorted(glob.glob("gpurun_out/ab/nw*_2.json")) + sorted(glob.glob("gpurun_out/ab/nw*_1.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "ms_per_step", round(d["ms_per_step"], 3), "LARGE ms", r.get("kernel_ms"), "frac", r.get("frac"))
    except Exception as e: print(f, "unreadable", e)
PY
mkdir -p gpurun_out/final_r06
python bench.py 2> gpurun_out/final_r06/default.err | grep '^{"metric"' | tail -1 > gpurun_out/final_r06/default.json
python -c "import json; d=json.load(open('gpurun_out/final_r06/default.json')); r=d['roofline']; print('default', d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r.get('rocprof_ms'), r.get('rocprof_frac'), d.get('verified'), (d.get('verified_vs_reference') or {}).get('ok'))"
  ;;
29)
# round 6, twenty-ninth session: w4s_segment in two launches - the 64-lead instance (82 registers, 3.6 KB of LDS: six waves per SIMD, two
# unrolled rounds) over every block, the large instance over the list of blocks that own a window of more than 64 leads.  GPU parity of
# the front end, stamps, same-box A/B against one launch (SNF_W4_SPLIT=0), the front end's timeline entries
timeout 900 python -m pytest tests/test_prefilter.py tests/test_gpu_parity.py tests/test_zz_gpu_end_to_end.py -m gpu -x -q 2>&1 | tail -1
bash tools/itrace.sh 2>&1 | grep -E "in flight$|w4s_segment"
bash tools/run_ab.sh -n 3 one:SNF_W4_SPLIT=0 two: 2>&1 | tee gpurun_out/ab_r06_19.log
for sp in 0 1; do SNF_W4_SPLIT=$sp SNF_PROF=1 SNF_TIMELINE=1 python bench.py $Q --no-verify --steps 3 --warmup 2 --inflight 1 2>&1 | grep -E "SNF_TIMELINE.*(w4s_segment|w3_scatter|w5|w6t_emit)|window front end" | tail -5; done 2>&1 | tee -a gpurun_out/ab_r06_19.log
  ;;
esac
