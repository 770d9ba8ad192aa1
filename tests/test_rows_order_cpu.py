"""The launch order of the 384-wide diffusion sweep (greb_rows.hip: rows_tasks) is a SPEED choice -- interleaved chain
and streaming strips, ever shorter strips for the fields launched last -- but it must be a partition: every latitude
row of every field is written by exactly one task.  Host-only entry point, no GPU needed."""
import numpy as np
import pytest

from greb_climate_model_amd import abi, engine


@pytest.mark.parametrize("batch", [1, 2, 7, 8, 9, 37, 1024])
@pytest.mark.parametrize("kappa", [None, 7.2e5])
def test_every_row_of_every_field_exactly_once(batch, kappa):
    p = abi.default_params()
    if kappa is not None:
        p.kappa = kappa  # the two polar rows become 1 800-sweep chains: another cut of the caps
    field, k0, k1, up = engine.diffusion_launch_order(p, 384, 192, batch)
    assert len(field) % 8 == 0 and len(field) > 0
    live = field >= 0
    assert field[live].max() == batch - 1 and (field[~live] == -1).all()
    cover = np.zeros((batch, 192), np.int32)
    for f, a, b in zip(field[live], k0[live], k1[live]):
        assert 0 <= a < b <= 192
        cover[f, a:b] += 1
    assert (cover == 1).all()
    # groups of eight: the same strip of eight consecutive fields (blocks are dealt to the eight XCDs in turn)
    g = np.arange(len(field)) // 8
    for arr in (k0, k1, up):
        assert (arr == arr[g * 8]).all()
    f0 = field[::8]
    assert (f0 % 8 == 0).all()


def test_last_tasks_are_the_short_streaming_strips():
    p = abi.default_params()
    field, k0, k1, up = engine.diffusion_launch_order(p, 384, 192, 1024)
    rows = (k1 - k0)[field >= 0]
    assert rows[-512:].max() <= 8 < rows[:2048].max()
    # neighbouring streaming strips of a field walk away from their common border
    f = 5
    mine = sorted((a, b, u) for ff, a, b, u in zip(field, k0, k1, up) if ff == f and b - a > 15)
    assert len(mine) >= 2 and all(x[2] != y[2] for x, y in zip(mine, mine[1:]))


def test_other_grids_keep_the_band_kernel():
    for nx, ny in ((96, 48), (192, 96)):
        assert len(engine.diffusion_launch_order(abi.default_params(), nx, ny, 4)[0]) == 0


@pytest.mark.parametrize("n_members", [1, 3, 8, 40, 62])
def test_substep_order_is_a_partition_in_one_round(n_members):
    """The engine's row-strip sub-step (greb_step_rows.hip: step_rows_tasks): every row of every (member, tracer) field
    exactly once -- with per-member diffusivities, i.e. different sub-cycle tables per member --, never more tasks than
    the chip has wavefront slots (2 048: a task started late ends the launch late), and the two tasks that share a SIMD
    (i and i + 1 024) never both hold a 232-sweep polar row; with a SIMD per task the dearest strips lead the launch."""
    p = abi.default_params()
    kappa = np.float32(8e5) * (1 + 0.05 * np.sin(np.arange(n_members)))  # stays above 7.27e5: no 1 800-sweep polar rows
    field, k0, k1 = engine.substep_launch_order(p, 384, 192, n_members, kappa.astype(np.float32))
    assert field.min() == 0 and field.max() == 2 * n_members - 1
    cover = np.zeros((2 * n_members, 192), np.int32)
    for f, a, b in zip(field, k0, k1):
        assert 0 <= a < b <= 192
        cover[f, a:b] += 1
    assert (cover == 1).all()
    n = len(field)
    assert n <= 2048
    long_chain = np.array([(a <= 1 < b) or (a <= 190 < b) for a, b in zip(k0, k1)])  # rows 1 and 190: 232 sweeps
    assert long_chain.sum() == 4 * n_members
    if n <= 1024:
        assert long_chain[: 4 * n_members].all()
    else:
        assert n > 1900  # the slots are used
        assert not (long_chain[: n - 1024] & long_chain[1024:]).any()
    assert len(engine.substep_launch_order(p, 96, 48, 2)[0]) == 0  # other grids keep the band kernels


@pytest.mark.parametrize("n_members,slots", [(1, 2048), (3, 2048), (8, 2048), (40, 2048), (62, 2048), (62, 1984), (2, 64), (5, 200)])
def test_circulation_plan_is_a_partition_with_every_dependency_in_it(n_members, slots):
    """The one-launch circulation call (greb_circ_rows.hip: circ_rows_tasks; src/greb.f90:546-550): every row of every
    field is owned by exactly one task; there are never more tasks than the wavefront slots given (the launch waits inside
    the kernel for its own tasks: all of them must be resident at once); rows with >= 64 dependent diffusion sweeps are
    chain tasks of one row; and the dependency table is complete and minimal: a task lists exactly the owners of the rows
    k0-2, k0-1, k1, k1+1 of ITS field that are not itself -- each exists, each once, all in the same launch."""
    p = abi.default_params()
    kappa = (np.float32(8e5) * (1 + 0.05 * np.sin(np.arange(n_members)))).astype(np.float32)
    if n_members == 2:
        kappa[:] = 7.2e5  # config 5's second engine: 1 800-sweep polar rows (src/greb.f90:652-654)
    field, k0, k1, chain, dep = engine.circulation_launch_plan(p, 384, 192, n_members, kappa, slots)
    n = len(field)
    assert 0 < n <= slots
    owner = -np.ones((2 * n_members, 192), np.int64)
    for i, (f, a, b) in enumerate(zip(field, k0, k1)):
        assert 0 <= a < b <= 192 and 0 <= f < 2 * n_members
        assert (owner[f, a:b] == -1).all()
        owner[f, a:b] = i
    assert (owner >= 0).all()
    for i in range(n):
        want = set()
        for k in (k0[i] - 2, k0[i] - 1, k1[i], k1[i] + 1):
            if 0 <= k < 192 and owner[field[i], k] != i:
                want.add(int(owner[field[i], k]))
        got = [int(d) for d in dep[i] if d >= 0]
        assert len(got) == len(set(got)) and set(got) == want, (i, got, want)
        assert all(field[d] == field[i] for d in got)
        if chain[i]:
            assert k1[i] == k0[i] + 1
    # which rows are chain tasks follows from the row's own sub-cycle count alone, the same in every field of a table
    sweeps = {1: 225, 2: 82, 189: 82, 190: 225}
    for f in range(2 * n_members):
        rows = sorted(int(k0[i]) for i in range(n) if field[i] == f and chain[i])
        if n_members != 2:
            assert rows == sorted(sweeps), (f, rows)
        else:
            assert rows == [0, 1, 2, 189, 190, 191], (f, rows)


def test_circulation_plan_declines_what_cannot_be_resident():
    p = abi.default_params()
    assert len(engine.circulation_launch_plan(p, 384, 192, 62, None, 300)[0]) == 0   # 124 fields x >= 5 tasks
    assert len(engine.circulation_launch_plan(p, 96, 48, 2, None, 2048)[0]) == 0     # other grids: other kernels
