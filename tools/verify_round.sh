#!/bin/bash
# End-of-round checklist (build container).  Each GPU step is one gpurun call; nothing runs in parallel.
#   tools/verify_round.sh            CPU part only
#   tools/verify_round.sh gpu        + GPU tests, smoke, bench on a MI355X box (about 4 GPU-minutes)
#   tools/verify_round.sh profiles   + rocprofv3 kernel stats and PMC passes, copied into profiles/ (about 3 more)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
python -m pytest tests -x -q -m "not gpu"
[ "$1" = gpu ] || [ "$1" = profiles ] || exit 0
G=/usr/local/graft/bin/gpurun
$G --timeout 1100 -- 'python -m pytest tests -m gpu -x -q 2>&1 | tail -3; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; python bench.py > gpurun_out/bench_final.log 2>&1; grep "^{" gpurun_out/bench_final.log | cut -c1-220'
grep '^{' gpurun_out/bench_final.log | tail -1 > profiles/r01_bench_line.json
[ "$1" = profiles ] || exit 0
$G --timeout 1100 -- 'R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round -o bench -- python3 $R/bench.py > $R/gpurun_out/prof_round_bench.log 2>&1; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_round_fetch -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_round_write -o runc -- python3 $R/tools/microbench_dif.py 16384 > /dev/null 2>&1; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_round_sq -o runc -- python3 $R/bench.py --no-cpu --no-roofline --steps 2 > /dev/null 2>&1; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_round_lds -o runc -- python3 $R/bench.py --no-cpu --no-roofline --steps 2 > /dev/null 2>&1; echo done'
cp gpurun_out/prof_round/bench_kernel_stats.csv profiles/r01_bench_kernel_stats.csv
grep '^{' gpurun_out/prof_round_bench.log | tail -1 > profiles/r01_bench_under_rocprof.json
python tools/pmc_summary.py gpurun_out/pmc_round_fetch/runc_counter_collection.csv gpurun_out/pmc_round_write/runc_counter_collection.csv > profiles/r01_diffusion_pmc.txt
python tools/pmc_summary.py gpurun_out/pmc_round_sq/runc_counter_collection.csv gpurun_out/pmc_round_lds/runc_counter_collection.csv > profiles/r01_member_sq_pmc.txt
echo "profiles/ refreshed; update profiles/r01_roofline_traffic.json from r01_diffusion_pmc.txt if the kernel's traffic changed"
