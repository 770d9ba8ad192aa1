#!/bin/bash
# Member-kernel time split (GPU box): bench at nsub = 24 and 48 sub-steps per model step (GREB_DEBUG_NSUB; the
# nsub=48 results are physically meaningless, timing only).  slope = (t48 - t24)/24 ~ one circulation sub-step,
# t24 - 24*slope ~ everything else in a model step (wind staging, point physics, accumulation, barriers).
for ns in 24 48; do
  GREB_DEBUG_NSUB=$ns python bench.py --no-cpu --no-roofline --steps 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('nsub=$ns', d['value'], 'yr/s', d['ms_per_step'], 'ms/yr ->', round(d['ms_per_step']/2/730*1e3, 2), 'us per member-step')"
done
