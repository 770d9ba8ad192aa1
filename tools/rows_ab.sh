#!/bin/bash
# A/B timing of the 384x192 row-strip diffusion kernel under the -DGREB_TUNING knobs (one gpurun call).
cd "$(dirname "$0")/.."
export REPS=8
run() { echo "== $*"; env "$@" python tools/microbench_dif.py ${B:-1024} ${G:-384 192} 2>&1 | grep "strict=False"; }
run A=0
run GREB_DEBUG_ROWS=1
for n in 2 4 6; do
run GREB_LIB=variants/libgreb_s$n.so
run GREB_LIB=variants/libgreb_s$n.so GREB_DEBUG_ROWS=1
done
run GREB_LIB=variants/libgreb_s6.so GREB_ROWS_LDS_PAD=6000
run GREB_LIB=variants/libgreb_s4.so GREB_ROWS_LDS_PAD=4000
