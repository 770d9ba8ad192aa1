#!/bin/bash
# End-of-round checklist (build container).  Each GPU step is one gpurun call; nothing runs in parallel.
#   tools/verify_round.sh            CPU part only
#   tools/verify_round.sh gpu        + GPU tests, smoke, bench on a MI355X box (about 6 GPU-minutes)
#   tools/verify_round.sh profiles   + rocprofv3 kernel stats and PMC passes, copied into profiles/ (about 7 more)
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r03}
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
python -m pytest tests -x -q -m "not gpu"
[ "$1" = gpu ] || [ "$1" = profiles ] || exit 0
G=/usr/local/graft/bin/gpurun
$G --timeout 1100 -- 'python -m pytest tests -m gpu -x -q -s > gpurun_out/gputest_final.log 2>&1; tail -3 gpurun_out/gputest_final.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; python bench.py > gpurun_out/bench_final.log 2>&1; grep "^{" gpurun_out/bench_final.log | cut -c1-220'
grep '^{' gpurun_out/bench_final.log | tail -1 > profiles/${R}_bench_line.json
grep -E "rms|zonal|passed|failed|global mean" gpurun_out/gputest_final.log > profiles/${R}_gpu_parity_numbers.txt || true
[ "$1" = profiles ] || exit 0
# kernel trace of the bench command; PMC passes on their own (never combined with a trace)
$G --timeout 1100 -- 'R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/prof_round_bench.log 2>&1; export WARM=20 REPS=1; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_round_fetch -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_round_write -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_round_sq -o runc -- python3 $R/bench.py --no-cpu --no-roofline --no-g384 --steps 2 > /dev/null 2>&1; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_round_lds -o runc -- python3 $R/bench.py --no-cpu --no-roofline --no-g384 --steps 2 > /dev/null 2>&1; cd $R && python tools/stamp_member.py 512 2 > gpurun_out/stamp_round.log 2>&1; python tools/stamp_step_rows.py > gpurun_out/stamp_step_round.log 2>&1; echo done'
# the 384x192 diffusion sweep: the row-strip kernel and, for comparison, the band kernel it replaces (GREB_NO_ROWS=1 in
# the tuning library); the per-launch spread from an idle GPU at both grids
$G --timeout 1100 -- 'tools/prof_rows.sh rows > /dev/null 2>&1; GREB_NO_ROWS=1 tools/prof_rows.sh band > /dev/null 2>&1; python tools/launch_spread.py 96 48 16384 2>&1 | grep -v amdgpu > gpurun_out/spread_g96.txt; python tools/launch_spread.py 384 192 1024 2>&1 | grep -v amdgpu > gpurun_out/spread_g384.txt; echo done'
hipcc --offload-arch=gfx950 -O3 -I greb_climate_model_amd/csrc tools/ubench/chain_rate.hip -o tools/ubench/chain_rate
# the 384x192 engine's sub-step: counter passes at 62 members, the per-task timeline (slots, pairs, what ends the launch),
# its critical chain's stamps and the chain loops on a lone wavefront
$G --timeout 1100 -- 'tools/prof_step.sh 62 > /dev/null 2>&1; for m in 1 24 62; do python tools/step_timeline.py $m 2>&1 | grep -v amdgpu; echo; done > gpurun_out/timeline_round.txt; python tools/stamp_step_rows.py 62 2>&1 | grep -v amdgpu > gpurun_out/stamp_step62_round.log; tools/ubench/chain_rate > gpurun_out/chain_rate_round.txt 2>&1; echo done'
{ echo "# 384x192 sub-step kernel (greb_step_rows.hip), 62 perturbed-physics members = 124 fields: counter passes of tools/prof_step.sh"; echo "# (each rocprofv3 --pmc pass on its own, 16 model steps = 384 launches; SQ_* wave counters in quad-cycles; 'per field' = / 124)"; cat gpurun_out/r3_step62_pmc.txt; } > profiles/${R}_g384_substep_pmc.txt
cp gpurun_out/timeline_round.txt profiles/${R}_g384_substep_timeline.txt
cp gpurun_out/chain_rate_round.txt profiles/${R}_chain_rate.txt
cp gpurun_out/prof_round/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
grep '^{' gpurun_out/prof_round_bench.log | tail -1 > profiles/${R}_bench_under_rocprof.json
python tools/pmc_summary.py gpurun_out/pmc_round_fetch/runc_counter_collection.csv gpurun_out/pmc_round_write/runc_counter_collection.csv > profiles/${R}_diffusion_pmc.txt
python tools/pmc_summary.py --traffic-json gpurun_out/pmc_round_fetch/runc_counter_collection.csv gpurun_out/pmc_round_write/runc_counter_collection.csv > profiles/${R}_roofline_traffic.json
python tools/pmc_summary.py gpurun_out/pmc_round_sq/runc_counter_collection.csv gpurun_out/pmc_round_lds/runc_counter_collection.csv > profiles/${R}_member_sq_pmc.txt
grep -v amdgpu.ids gpurun_out/stamp_round.log > profiles/${R}_member_stamps.txt
{ grep -v amdgpu.ids gpurun_out/stamp_step_round.log; echo; cat gpurun_out/stamp_step62_round.log; } > profiles/${R}_g384_substep_stamps.txt
python tools/pmc_summary.py --derive gpurun_out/pmc_round_sq/runc_counter_collection.csv gpurun_out/pmc_round_lds/runc_counter_collection.csv gpurun_out/prof_round/bench_kernel_stats.csv >> profiles/${R}_member_sq_pmc.txt || true
{ echo "# 384x192 batched diffusion sweep, batch 1 024 (906 MB algorithmic per launch): counter passes of tools/prof_rows.sh"; echo "# (each rocprofv3 --pmc pass on its own; short runs: the kernel-trace durations are those of unsettled clocks,"; echo "#  the settled per-launch times are in ${R}_bench_line.json: g384.diffusion_sweep)"; echo; echo "######## the row-strip kernel (greb_rows.hip), this round"; cat gpurun_out/r3_rows_pmc.txt; echo; echo "######## the band kernel it replaces (greb_kernels.hip: sweep_kernel<.,dif>; GREB_NO_ROWS=1 in the tuning library)"; cat gpurun_out/r3_band_pmc.txt; } > profiles/${R}_g384_diffusion_pmc.txt
cp gpurun_out/r3_rows_pmc.json profiles/${R}_g384_diffusion_pmc.json
cat gpurun_out/spread_g96.txt gpurun_out/spread_g384.txt > profiles/${R}_launch_spread.txt
echo "profiles/ refreshed"
