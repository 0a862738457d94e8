#!/bin/bash
# tools/r05_sessions.sh N: the GPU sessions of round 5 (one `gpurun -- bash tools/r05_sessions.sh N` each), in the order they were run;
# what they wrote is under profiles/ (profiles/README.md, "Round 5").  Sessions 1-5 and 8-15 are same-box A/Bs and probes, 6 is the profile
# set of the round (run again on the final sources), 7 what the driver runs at round end, 16 the population merge over its runs of tasks.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
case "$1" in
1)
# round 5, first GPU session: the GPU suite on the new front end (w4s_segment / w6t_emit) and d1g_refine<8>, a same-box A/B against
# round 4's library (variants/base_r04.so), the launch timeline
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: nod1g:SNF_NO_D1_GROUPS=1 div2:SNF_GRID_DIV=2 2>&1 | tee gpurun_out/ab_r05_1.log
bash tools/timeline1.sh > /dev/null 2>&1; head -60 gpurun_out/timeline1.txt
  ;;
2)
# round 5, second GPU session: w4s_segment with tagged keys, the result staged through HBM (SNF_STAGE_OUT=1), f4w_emit on a small grid,
# SQ counters of the new kernels, the per-task split
timeout 600 python -m pytest tests/test_prefilter.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu2.log 2>&1; tail -3 gpurun_out/pytest_gpu2.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: stage:SNF_STAGE_OUT=1 f4:SNF_F4_GRID=256 nod1g:SNF_NO_D1_GROUPS=1 2>&1 | tee gpurun_out/ab_r05_2.log
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 40 --warmup 5"
for k in 1 2; do
  SNF_STAGE_OUT=1 $B --inflight 3 2>/dev/null | python -c "import json,sys; print('stage, three in flight', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"
  $B --inflight 3 2>/dev/null | python -c "import json,sys; print('new, three in flight', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"
done 2>&1 | tee -a gpurun_out/ab_r05_2.log
bash tools/sq_all.sh > /dev/null 2>&1; head -40 gpurun_out/sq_all/summary.txt
SNF_STAGE_OUT=1 bash tools/timeline1.sh > /dev/null 2>&1; cp gpurun_out/timeline1.txt gpurun_out/timeline1_stage.txt; head -50 gpurun_out/timeline1_stage.txt
timeout 300 python tools/per_task_prof.py 2>&1 | tail -4
  ;;
3)
# round 5, third GPU session: the staged result path as the library's rule (another pass in flight -> through HBM), the GPU tests that
# changed, a same-box A/B, the whole default bench line with the new legs, phase stamps of d1w_refine
timeout 900 python -m pytest tests/test_output_modes.py tests/test_insitu_seam.py tests/test_dropin_api.py tests/test_prefilter.py -m gpu -x -q > gpurun_out/pytest_gpu3.log 2>&1; tail -3 gpurun_out/pytest_gpu3.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: direct:SNF_STAGE_OUT=0 order0:SNF_CONS_ORDER=0 2>&1 | tee gpurun_out/ab_r05_3.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_3.json 2> gpurun_out/bench_default_3.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_3.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"))
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
print("workers", json.dumps(wc.get("worker_processes")))
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "records_compared", "vs_reference_all_cores", "reference_leg_s", "seconds", "error", "reference_error")})
print("roofline", d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
print([(k["name"], k["ms"]) for k in d["roofline"]["top_kernels"]])
PY
tail -5 gpurun_out/bench_default_3.err
SNF_LIB_SO=$R/variants/prof.so SNF_PROF=1 timeout 300 python bench.py --inflight 1 --no-cpu-baseline --no-wall-clock --no-configs --no-verify --steps 3 --warmup 1 2>&1 | grep "CONS_PROFILE\] d1w" | tail -6
  ;;
4)
# round 5, fourth GPU session: the staged result with the copies at the fetch (default) and as copy kernels inside the pass, the consensus
# order under it, the per-task split with the library's own upload split, worker processes with a bounded number of hardware queues
timeout 900 python -m pytest tests/test_output_modes.py tests/test_dropin_api.py tests/test_abi.py -m gpu -x -q > gpurun_out/pytest_gpu4.log 2>&1; tail -3 gpurun_out/pytest_gpu4.log
bash tools/run_ab.sh -n 2 base:SNF_LIB_SO=$R/variants/base_r04.so new: kcopy:SNF_STAGE_COPY=kernel order0:SNF_CONS_ORDER=0 kcopy0:SNF_STAGE_COPY=kernel,SNF_CONS_ORDER=0 2>&1 | tee gpurun_out/ab_r05_4.log
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -8
timeout 600 python tools/bench_workers.py 4 8 24 q0 q2 2>&1 | grep -v "^\[" | cut -c1-260 | tee gpurun_out/workers_4.log
  ;;
5)
# round 5, fifth GPU session: the whole GPU suite, the default line as the driver runs it (all legs), the upload split per task
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; tail -3 gpurun_out/pytest_gpu5.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_5.json 2> gpurun_out/bench_default_5.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_5.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"), "one in flight", d["config"]["ms_per_pass_one_batch_in_flight"])
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")))
for k, v in (wc.get("worker_processes") or {}).items():
    print("workers", k, v if not isinstance(v, dict) else {x: v[x] for x in ("hot_all_ms", "ingest_all_ms", "hw_queues_per_process", "n_out")})
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "records_compared", "vs_reference_all_cores", "seconds", "error", "reference_error")})
r = d["roofline"]
print("roofline", r["kernel"], r["kernel_ms"], r["frac"], "| stage", r.get("dominant_stage"))
print([(k["name"], k["ms"]) for k in r["top_kernels"]])
PY
tail -3 gpurun_out/bench_default_5.err
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -5
  ;;
6)
# round 5, sixth GPU session: the profile set of the round on the final kernel sources (tools/profile.sh), SQ counters incl. LDS bank
# conflicts (tools/sq_all.sh), the bench lines kept for judging (tools/final_set.sh), HBM traffic at four genomes per batch
bash tools/profile.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -45 gpurun_out/profile_r05.log
bash tools/sq_all.sh > /dev/null 2>&1; head -30 gpurun_out/sq_all/summary.txt | cut -c1-210
bash tools/final_set.sh r05 2>&1 | tail -12
bash tools/pmc.sh --genomes 4 > gpurun_out/pmc_g4.log 2>&1; cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_genomes4.json 2>/dev/null; tail -3 gpurun_out/pmc_g4.log
  ;;
7)
# round 5, seventh GPU session: what the driver runs at round end - the GPU suite, smoke(), the default line - on the final tree,
# and the per-task upload split behind the cached genotype table
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu7.log 2>&1; tail -3 gpurun_out/pytest_gpu7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_7.json 2> gpurun_out/bench_default_7.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_7.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "verified", d.get("verified"), d.get("verified_vs_reference"), "one in flight", d["config"]["ms_per_pass_one_batch_in_flight"])
print("vs_baseline", json.dumps(d["cpu_baseline"].get("vs_baseline")))
wc = d.get("wall_clock", {})
print("per_task", json.dumps(wc.get("per_task_api")), json.dumps(wc.get("per_task_execute")), wc["batched"]["end_to_end_ms"])
for k, v in d.get("configs", {}).items():
    print("config", k, {x: v.get(x) for x in ("ms_per_step", "verified", "verified_vs_reference", "seconds", "error", "reference_error")})
print("config 4 baseline", json.dumps(d["configs"]["4"].get("cpu_baseline"))[:900])
r = d["roofline"]
print("roofline", r["kernel"], r["kernel_ms"], r["frac"], r.get("rocprof_ms"), r.get("rocprof_frac"), "| stage", r.get("dominant_stage"))
print("issue", r.get("issue"))
PY
timeout 300 python tools/per_task_prof.py prof 2>&1 | grep -E "round|upload:" | tail -4
  ;;
8)
# round 5, eighth GPU session: knobs whose optimum may have moved with the staged result and the segment front end
bash tools/run_ab.sh -n 2 new: w9:SNF_WIN_BITS=9 w8:SNF_WIN_BITS=8 nw4:SNF_CONS_NW=4 nw4o0:SNF_CONS_NW=4,SNF_CONS_ORDER=0 mid:SNF_D2_MID=1 2>&1 | tee gpurun_out/ab_r05_5.log
  ;;
9)
# round 5, ninth GPU session: how often do two passes in flight fall into step?  Ten runs of the headline with the pacing rules, ten without
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8 9 10; do
  for p in 1 0; do
    SNF_PACE=$p $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pace $p run $i: ms_per_step %.3f  LARGE %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done 2>&1 | tee gpurun_out/pace_r05.log
  ;;
10)
# round 5, tenth GPU session: which pacing rule keeps two passes out of step at what price (eight runs each)
B="python bench.py --no-cpu-baseline --no-wall-clock --no-configs --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8; do
  for v in "SNF_PACE=2" "SNF_PACE=3" "SNF_PACE=1 SNF_PACE_FRAC=0.25" "SNF_PACE=3 SNF_PACE_FRAC=0.25"; do
    env $v $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v run $i: ms_per_step %.3f' % d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/pace2_r05.log
  ;;
11)
# round 5, eleventh GPU session: list-fed wave kernels on larger grids (fewer clusters per wave, more waves in the dispatcher's hands)
bash tools/run_ab.sh -n 2 new: gm2:SNF_GRID_MULT=2 gm4:SNF_GRID_MULT=4 2>&1 | tee gpurun_out/ab_r05_6.log
SNF_GRID_MULT=4 bash tools/timeline1.sh > /dev/null 2>&1; grep -E "d1g|d1w|d2g|d2w|e1w|span" gpurun_out/timeline1.txt
bash tools/timeline1.sh > /dev/null 2>&1; grep -E "d1g|d1w|d2g|d2w|e1w|span" gpurun_out/timeline1.txt
  ;;
12)
# round 5, twelfth GPU session: the N > 1 result path (shared landing) with one rank: staged against direct stores
Q="--no-configs --no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms_per_step %.3f' % d['ms_per_step'], [(k['name'], k['ms']) for k in d['roofline']['top_kernels'][:6]])"; }
for i in 1 2; do
SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared auto"
SNF_BENCH_FORCE_DIST=1 SNF_STAGE_OUT=0 python bench.py --gpus 1 $Q 2>/dev/null | show "shared direct"
SNF_BENCH_FORCE_DIST=1 SNF_STAGE_OUT=1 SNF_TIME_ALL=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared staged, every kernel timed"
SNF_BENCH_RESMEM=shm python bench.py --gpus 1 $Q 2>/dev/null | show "plain, results into a /dev/shm segment"
python bench.py --gpus 1 $Q 2>/dev/null | show "plain"
done
  ;;
13)
# round 5, thirteenth GPU session: what costs the shared-landing path (the N > 1 result path, one rank) its 16 % and its spread
Q="--no-configs --no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms_per_step %.3f' % d['ms_per_step'], [(k['name'], k['ms']) for k in d['roofline']['top_kernels'][:4]])"; }
for i in 1 2 3; do
SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared"
SNF_BENCH_FORCE_DIST=1 SNF_PACE=0 python bench.py --gpus 1 $Q 2>/dev/null | show "shared, no pacing"
SNF_BENCH_FORCE_DIST=1 SNF_BENCH_SHARED_DEBUG=nogather python bench.py --gpus 1 $Q 2>/dev/null | show "shared, no gather"
SNF_BENCH_FORCE_DIST=1 SNF_BENCH_SHARED_DEBUG=nogather,oneslot,noset python bench.py --gpus 1 $Q 2>/dev/null | show "shared, no gather, one segment per handle set once"
done
;;
14)
# round 5, fourteenth GPU session: the shared-landing path with more hardware queues (RCCL's streams + eight of ours share four by default)
Q="--no-configs --no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms_per_step %.3f' % d['ms_per_step'], [(k['name'], k['ms']) for k in d['roofline']['top_kernels'][:4]])"; }
for i in 1 2; do
SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared"
GPU_MAX_HW_QUEUES=8 SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared, 8 hardware queues"
GPU_MAX_HW_QUEUES=16 SNF_BENCH_FORCE_DIST=1 python bench.py --gpus 1 $Q 2>/dev/null | show "shared, 16 hardware queues"
GPU_MAX_HW_QUEUES=8 python bench.py --gpus 1 $Q 2>/dev/null | show "plain, 8 hardware queues"
python bench.py --gpus 1 $Q 2>/dev/null | show "plain"
done
;;
15)
# round 5, fifteenth GPU session: the plain line over the number of hardware queues the runtime may open (default 4: two handles' eight streams share them)
Q="--no-configs --no-cpu-baseline --no-wall-clock --steps 20 --warmup 5"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms_per_step %.3f' % d['ms_per_step'], [(k['name'], k['ms']) for k in d['roofline']['top_kernels'][:4]])"; }
for i in 1 2; do
for q in 2 3 4 5 6; do GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 $Q 2>/dev/null | show "plain, $q hardware queues"; done
done
;;
16)
# round 5, sixteenth GPU session: the host side of the population merge after its rework (records from arrays, whole-table assembly, one
# sort key, runs of tasks on two threads) over the number of runs; the per-task legs after the object materialiser's rework; the GPU
# tests the two touch
timeout 600 python tools/combine_host_prof.py --gpu --scale 1.0 --passes 4 --chunks 1,2,3,4,6,8 2>&1 | grep -v "^chunks [0-9]* pass [0-2]" | cut -c1-600 | tee gpurun_out/r05_merge_runs.log
( time timeout 900 python bench.py --config 4 --no-cpu-baseline --no-reference-baseline > gpurun_out/bench_config4_16.json 2> gpurun_out/bench_config4_16.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('gpurun_out/bench_config4_16.json').read().strip().splitlines()[-1]); print('config 4 ms_per_step', d['ms_per_step'], d['config']['host_phases_ms'], 'text equal', d['config'].get('text_equals_object_path'), 'verified', d.get('verified'))"
( time timeout 900 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 20 --warmup 5 > gpurun_out/bench_wall_16.json 2> gpurun_out/bench_wall_16.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('gpurun_out/bench_wall_16.json').read().strip().splitlines()[-1]); w=d['wall_clock']; print('ms_per_step', d['ms_per_step']); print(json.dumps({k: w[k] for k in w if k.startswith('per_task') or k in ('batched',)})[:3000])"
timeout 900 python -m pytest tests/test_dropin_api.py tests/test_insitu_seam.py tests/test_combine_task.py tests/test_pipeline.py tests/test_reference_pool.py -m gpu -x -q 2>&1 | tail -3
;;
*) echo "usage: bash tools/r05_sessions.sh 1..16"; exit 2 ;;
esac
