#!/bin/bash
# A/B timing of the 384x192 row-strip diffusion kernel under the -DGREB_TUNING knobs (one gpurun call).
cd "$(dirname "$0")/.."
export REPS=8
run() { echo "== $*"; env GREB_TUNING_LIB=1 "$@" python tools/microbench_dif.py ${B:-1024} ${G:-384 192} 2>&1 | grep "strict=False"; }
run A=0
run GREB_DEBUG_ROWS=4
run GREB_ROWS_LDS_PAD=900
run GREB_ROWS_LDS_PAD=1800
run A=0
run GREB_ROWS_SPAN=75
run GREB_ROWS_SPAN=65
run GREB_ROWS_L2=3072 GREB_ROWS_L3=1536
