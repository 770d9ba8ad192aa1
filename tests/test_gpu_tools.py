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


def test_a_circulation_launch_that_cannot_complete_is_an_error_not_a_hang():
    """The one-launch circulation call (greb_circ_rows.hip) waits inside the kernel for its own tasks; every such wait is
    bounded.  The tuning library can make one task leave without publishing (GREB_CIRC_LOSE_TASK) -- what a launch that is
    not resident as a whole looks like to its neighbours -- with the bound shortened to 20 ms (GREB_CIRC_SPIN_MS): the run
    must come back within seconds with GREB_E_STATE and the task and sub-step it gave up on, the abort must be sticky (the
    year's remaining launches fall through), and the process must exit cleanly."""
    from greb_climate_model_amd import build
    if not os.path.exists(build.LIB_TUNING):
        pytest.skip("no tuning library in this tree")
    code = """
import sys, time
sys.path.insert(0, %r)
from greb_climate_model_amd import engine, workload
engine.use_tuning_build()
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
e = engine.Engine(inp, p, n_members=2, persistent=True)
t = time.perf_counter()
try:
    e.run(1, 680.0)
    print("NO ERROR")
except engine.GrebError as err:
    print("ERROR", err.code, "|", err, "| %%.2f s" %% (time.perf_counter() - t))
e.close()
print("closed")
""" % ROOT
    env = dict(os.environ, GREB_CIRC_LOSE_TASK="5", GREB_CIRC_SPIN_MS="20")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "closed" in r.stdout, out[-2000:]
    m = re.search(r"ERROR (-?\d+) \| (.*) \| ([0-9.]+) s", r.stdout)
    assert m, out[-2000:]
    assert int(m.group(1)) == -3 and "given up" in m.group(2) and "sub-step" in m.group(2), m.group(2)  # GREB_E_STATE
    assert float(m.group(3)) < 30.0, m.group(3)  # one bounded wait + 729 launches that fall through, not 730 x 24 waits


def test_the_one_launch_circulation_pairs_the_tasks_its_order_means_to_pair():
    """circ_rows_tasks pairs the two tasks of a SIMD ("dearest with cheapest").  The one-launch kernel runs four tasks per
    workgroup, one per SIMD of the workgroup's compute unit; what was OBSERVED and is relied on for speed only: the second
    workgroup on a compute unit starts one SIMD further on, so task 4c + w shares its SIMD with task n_simd + 4c + (w + 3)
    mod 4 -- the order is built for exactly that.  Read from HW_REG_HW_ID / XCC_ID per task (tools/circ_timeline.py)."""
    from greb_climate_model_amd import build
    if not os.path.exists(build.LIB_TUNING):
        pytest.skip("no tuning library in this tree")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "circ_timeline.py"), "62", "8"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    print(r.stdout)
    m = re.search(r"PREMISE task 4c \+ w shares its SIMD with task (\d+) \+ 4c \+ \(w \+ 3\) mod 4: (\d+) of (\d+) SIMDs", r.stdout)
    g = re.search(r"workgroups whose four tasks sit on the four SIMDs of one CU: (\d+) of (\d+)", r.stdout)
    assert m and g, r.stdout[-2000:]
    n_simd, hit, pairs = map(int, m.groups())
    assert n_simd == 1024 and pairs > 900 and hit >= 0.97 * pairs, (n_simd, hit, pairs)
    assert int(g.group(1)) >= 0.97 * int(g.group(2)), g.groups()
