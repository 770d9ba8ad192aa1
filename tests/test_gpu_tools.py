"""GPU: premises the launch orders rest on, checked where they can be observed -- the tuning library (the release
library carries no stamp code), in a child process of its own."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workgroup_i_and_i_plus_n_simd_share_a_simd_on_an_idle_device():
    """step_rows_tasks / circ_rows_tasks pair the two tasks of a SIMD ("dearest with cheapest") on the OBSERVATION that
    workgroup i of a launch of single-wavefront workgroups lands on SIMD i mod n_simd when the device is otherwise idle
    (greb_step_rows.hip).  Performance only, never correctness -- but if the dispatcher stops doing it the pairing is
    dealt blind and nothing else would say so.  Read from HW_REG_HW_ID / XCC_ID per task (tools/step_timeline.py)."""
    from greb_climate_model_amd import build
    if not os.path.exists(build.LIB_TUNING):
        pytest.skip("no tuning library in this tree")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), "62"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    m = re.search(r"PREMISE tasks i and i \+ (\d+) share a SIMD: (\d+) of (\d+) SIMDs", r.stdout)
    assert m, r.stdout[-2000:]
    n_simd, hit, pairs = map(int, m.groups())
    print(r.stdout)
    assert n_simd == 1024 and pairs > 900, (n_simd, pairs)      # an MI355X, and the 62-member launch fills its SIMDs in pairs
    assert hit >= 0.97 * pairs, f"only {hit} of {pairs} SIMD pairs are (i, i + {n_simd}): the dispatch order the launch orders assume does not hold"
