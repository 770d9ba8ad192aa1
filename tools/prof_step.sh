#!/bin/bash
# Counter passes of the 384x192 engine's circulation kernels (circ_rows_kernel: one launch per call; step_rows_kernel: one
# per sub-step -- the engine's trial runs both) over one model year on the RELEASE library (the code whose hash the summary
# records): each rocprofv3 --pmc pass on its own, the program directly after `--`.
# Run on the GPU box: tools/prof_step.sh [members] [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-62}
TAG=${2:-step$M}
O=$R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/lds -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcc -o run -- python3 $R/tools/prof_g384.py $M > /dev/null 2>&1
cd $R && FIELDS=$((2 * M)) STEP_JSON=$R/gpurun_out/r4_${TAG}_pmc.json python3 tools/pmc_rows_summary.py $O > gpurun_out/r4_${TAG}_pmc.txt 2>&1; cat gpurun_out/r4_${TAG}_pmc.txt
