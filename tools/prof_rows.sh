#!/bin/bash
# Counter passes of the batched diffusion sweep at 384x192, batch 1 024 (each rocprofv3 --pmc pass on its own, the
# program directly after `--`), summarised into gpurun_out/r4_rows_pmc.txt.  Run on the GPU box: tools/prof_rows.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rows}
O=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp WARM=20 REPS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/lds -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcc -o run -- python3 $R/tools/microbench_dif.py 1024 384 192 > /dev/null 2>&1
cd $R && python3 tools/pmc_rows_summary.py $O gpurun_out/r4_${TAG}_pmc.json > gpurun_out/r4_${TAG}_pmc.txt 2>&1; cat gpurun_out/r4_${TAG}_pmc.txt
