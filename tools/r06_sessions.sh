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
esac
