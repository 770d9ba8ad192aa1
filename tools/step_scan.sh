#!/bin/bash
# A/B of the sub-step's launch-order knobs (tuning build): M=62 tools/step_scan.sh "ENV=.. ENV=.." "..."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
M=${M:-62}
for cfg in "$@"; do echo "== $cfg"; env $cfg python3 tools/g384_ab.py $M 2>&1 | grep members; done
