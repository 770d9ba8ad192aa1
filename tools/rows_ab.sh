#!/bin/bash
# A/B timing of the 384x192 row-strip diffusion kernel under the -DGREB_TUNING knobs (one gpurun call).
cd "$(dirname "$0")/.."
export REPS=8 GREB_ROWS_P0=${GREB_ROWS_P0:-8}
run() { echo "== $*"; env "$@" python tools/microbench_dif.py ${B:-1024} 384 192 2>&1 | grep "strict=False"; }
run A=0
run GREB_ROWS_NT=1
run A=0
run GREB_ROWS_NT=1
run GREB_ROWS_P0=10 GREB_ROWS_P1=10
run GREB_ROWS_P0=12 GREB_ROWS_P1=12
run GREB_ROWS_SPAN=60
run GREB_ROWS_SPAN=50
run GREB_ROWS_L2=4096 GREB_ROWS_L3=2048
run GREB_ROWS_L2=1024 GREB_ROWS_L3=1024
run GREB_ROWS_L2=0 GREB_ROWS_L3=0
run GREB_ROWS_TARGET=6000
run GREB_ROWS_TARGET=2500
run GREB_DEBUG_ROWS=1
