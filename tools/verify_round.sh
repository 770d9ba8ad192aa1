#!/bin/bash
# End-of-round checklist (build container).  Each GPU step is one gpurun call; nothing runs in parallel.
#   tools/verify_round.sh            CPU part only
#   tools/verify_round.sh gpu        + GPU tests, smoke, bench on a MI355X box (about 8 GPU-minutes)
#   tools/verify_round.sh profiles   + rocprofv3 kernel stats, PMC passes, timelines and the full-length configs, copied into
#                                      profiles/ (about 12 more).  The counter passes run the RELEASE library and their
#                                      summaries record the hash of the profiled kernels' code (codesha.py): run this AFTER the
#                                      last kernel edit, or tests/test_profiles_cpu.py fails.
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r04}
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
if [ "$1" != profiles-only ] && [ -z "$SKIP_CPU" ]; then python -m pytest tests -x -q -m "not gpu" --deselect tests/test_profiles_cpu.py; fi
[ "$1" = gpu ] || [ "$1" = profiles ] || [ "$1" = profiles-only ] || exit 0
G=/usr/local/graft/bin/gpurun
if [ "$1" != profiles-only ]; then
$G --timeout 1190 -- 'python -m pytest tests -m gpu -x -q -s > gpurun_out/gputest_final.log 2>&1; tail -3 gpurun_out/gputest_final.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; python bench.py > gpurun_out/bench_final.log 2>&1; grep "^{" gpurun_out/bench_final.log | cut -c1-220'
grep '^{' gpurun_out/bench_final.log | tail -1 > profiles/${R}_bench_line.json
grep -E "rms|zonal|passed|failed|global mean|members .* persistent" gpurun_out/gputest_final.log > profiles/${R}_gpu_parity_numbers.txt || true
fi
[ "$1" = profiles ] || [ "$1" = profiles-only ] || exit 0
# (1) kernel trace of the bench command; the 96x48 roofline kernel's FETCH / WRITE passes; the fused member kernel's SQ passes
#     (every --pmc pass on its own, never combined with a trace; the program directly after `--`)
$G --timeout 1190 -- 'R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/prof_round_bench.log 2>&1; export WARM=20 REPS=1; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_round_fetch -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_round_write -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_round_sq -o runc -- python3 $R/bench.py --no-cpu --no-roofline --no-g384 --steps 2 > /dev/null 2>&1; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_round_lds -o runc -- python3 $R/bench.py --no-cpu --no-roofline --no-g384 --steps 2 > /dev/null 2>&1; cd $R && python tools/stamp_member.py 512 2 > gpurun_out/stamp_round.log 2>&1; echo done'
# (2) the 384x192 diffusion sweep (row strips, release library; the band kernel it replaces through the tuning library's
#     GREB_NO_ROWS); the engine's circulation kernels at 62 members; the one-launch call's per-task timeline; launch spread
$G --timeout 1190 -- 'tools/prof_rows.sh rows > /dev/null 2>&1; GREB_TUNING_LIB=1 GREB_NO_ROWS=1 tools/prof_rows.sh band > /dev/null 2>&1; tools/prof_step.sh 62 > /dev/null 2>&1; for m in 1 24 62; do python tools/circ_timeline.py $m 2>&1 | grep -v amdgpu; echo; done > gpurun_out/circ_timeline_round.txt; python tools/launch_spread.py 96 48 16384 2>&1 | grep -v amdgpu > gpurun_out/spread_g96.txt; python tools/launch_spread.py 384 192 1024 2>&1 | grep -v amdgpu > gpurun_out/spread_g384.txt; echo done'
# (3) both launch forms across member counts; 192x96; BASELINE configs 2-5 at full length on this library
$G --timeout 1190 -- '{ echo "# 384x192, perturbed-physics members (tools/g384_ab.py): member-yr/s and us per circulation sub-step (point-physics launch included)"; echo "one launch per call:     $(PERSISTENT=1 python tools/g384_ab.py 1 8 16 24 32 40 48 62 2>&1 | grep members | sed "s/member-yr.s, //; s/us per sub-step.*//; s/members //" | tr "\n" "|")"; echo "one launch per sub-step: $(PERSISTENT=0 python tools/g384_ab.py 1 8 16 24 32 40 48 62 2>&1 | grep members | sed "s/member-yr.s, //; s/us per sub-step.*//; s/members //" | tr "\n" "|")"; echo "# 192x96 (tools/bench_grid.py), default engine"; python tools/bench_grid.py 192 96 1 16 64 256 2>&1 | grep members | cut -c1-110; } > gpurun_out/forms_round.txt; for c in 2 3 4 5; do python tools/run_config.py $c 2>&1 | grep "^{"; done > gpurun_out/configs_round.jsonl; echo done'
cp gpurun_out/prof_round/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
grep '^{' gpurun_out/prof_round_bench.log | tail -1 > profiles/${R}_bench_under_rocprof.json
python tools/pmc_summary.py gpurun_out/pmc_round_fetch/runc_counter_collection.csv gpurun_out/pmc_round_write/runc_counter_collection.csv > profiles/${R}_diffusion_pmc.txt
python tools/pmc_summary.py --traffic-json gpurun_out/pmc_round_fetch/runc_counter_collection.csv gpurun_out/pmc_round_write/runc_counter_collection.csv > profiles/${R}_roofline_traffic.json
python tools/pmc_summary.py gpurun_out/pmc_round_sq/runc_counter_collection.csv gpurun_out/pmc_round_lds/runc_counter_collection.csv > profiles/${R}_member_sq_pmc.txt
python tools/pmc_summary.py --derive gpurun_out/pmc_round_sq/runc_counter_collection.csv gpurun_out/pmc_round_lds/runc_counter_collection.csv gpurun_out/prof_round/bench_kernel_stats.csv >> profiles/${R}_member_sq_pmc.txt || true
grep -v amdgpu.ids gpurun_out/stamp_round.log > profiles/${R}_member_stamps.txt
{ echo "# 384x192 batched diffusion sweep, batch 1 024 (906 MB algorithmic per launch): counter passes of tools/prof_rows.sh"; echo "# (each rocprofv3 --pmc pass on its own; short runs: the kernel-trace durations are those of unsettled clocks,"; echo "#  the settled per-launch times are in ${R}_bench_line.json: g384.diffusion_sweep)"; echo; echo "######## the row-strip kernel (greb_rows.hip), release library"; cat gpurun_out/r4_rows_pmc.txt; echo; echo "######## the band kernel it replaces (greb_kernels.hip: sweep_kernel<.,dif>; GREB_NO_ROWS=1 in the tuning library)"; cat gpurun_out/r4_band_pmc.txt; } > profiles/${R}_g384_diffusion_pmc.txt
cp gpurun_out/r4_rows_pmc.json profiles/${R}_g384_diffusion_pmc.json
{ echo "# 384x192 engine, 62 perturbed-physics members = 124 fields, one model year on the release library: counter passes of tools/prof_step.sh"; echo "# (each rocprofv3 --pmc pass on its own; circ_rows_kernel: one launch per circulation call = 24 sub-steps, the form the engine's"; echo "#  trial kept; step_rows_kernel: the 3 x 24 launches of that trial; SQ_* wave counters in quad-cycles; 'per field' = / 124)"; cat gpurun_out/r4_step62_pmc.txt; } > profiles/${R}_g384_substep_pmc.txt
cp gpurun_out/r4_step62_pmc.json profiles/${R}_g384_substep_pmc.json
{ echo "# The one-launch circulation call (greb_circ_rows.hip), tuning library: per task the time waiting for neighbours and, for chain"; echo "# tasks, the time inside the zonal chains, per sub-step (tools/circ_timeline.py; GREB_DEBUG_NSTEPS = 40 model steps)"; cat gpurun_out/circ_timeline_round.txt; } > profiles/${R}_circ_timeline.txt
cat gpurun_out/spread_g96.txt gpurun_out/spread_g384.txt > profiles/${R}_launch_spread.txt
cp gpurun_out/forms_round.txt profiles/${R}_launch_forms.txt
python - <<'EOF'
import json
rows = [json.loads(l) for l in open("gpurun_out/configs_round.jsonl") if l.startswith("{")]
json.dump({"source": "tools/run_config.py 2 / 3 / 4 / 5 at full length (3 + 50 / 50 / 100 / 50 yr) on one MI355X, this round's final library",
           "configs": rows}, open("profiles/r04_configs.json", "w"), indent=1)
EOF
python -m pytest tests/test_profiles_cpu.py -q
echo "profiles/ refreshed"
