"""The batched Nelder-Mead must make exactly the decisions of ``scipy.optimize.fmin`` (the
optimiser the reference calls, parrm.py:510-517,545-550): same iterates, same call counts."""

import numpy as np
import pytest
from scipy.optimize import fmin

from pyparrm_amd._neldermead import fmin_lockstep


def _objectives():
    def smooth(x):
        return (x - 3.3) ** 2 + 0.1 * np.sin(40 * x)

    def quantised(x):  # plateaus and ties: exercises contraction / shrink branches
        return np.round((x - 7.742) ** 2 * 50) / 50 + 1e-3 * np.abs(x)

    def spiky(x):
        return np.abs(np.sin(3 * x)) + 0.01 * (x - 2) ** 2

    def with_inf(x):
        return np.inf if x > 5.2 else (x - 5) ** 2

    def period_like(x):  # narrow minimum on a flat floor, like the PARRM objective
        return 1.0 - 0.6 * np.exp(-(((x - 169.2358) / 2e-3) ** 2))

    return [smooth, quantised, spiky, with_inf, period_like]


@pytest.mark.parametrize("fi", range(5))
def test_matches_scipy_fmin_single_start(fi):
    f = _objectives()[fi]
    starts = [0.7, 2.9, 5.0, 7.7, 169.2, 169.236, -3.0, 0.0, 1e-3]
    for x0 in starts:
        ref = fmin(lambda x: float(f(x[0])), x0, full_output=True, disp=False)
        got = fmin_lockstep(lambda xs: np.array([f(x) for x in xs]), [x0])[0]
        assert got[0].shape == (1,)
        assert got[0][0] == ref[0][0], (x0, got, ref[:4])
        assert got[1] == ref[1]
        assert got[2] == ref[2] and got[3] == ref[3]  # iterations, funcalls


def test_lockstep_equals_independent_runs_and_batches():
    f = _objectives()[0]
    starts = [0.7, 2.9, 3.31, 5.0, 10.0]
    batches = []

    def batch(xs):
        batches.append(len(xs))
        return np.array([f(x) for x in xs])

    together = fmin_lockstep(batch, starts)
    for x0, got in zip(starts, together):
        ref = fmin(lambda x: float(f(x[0])), x0, full_output=True, disp=False)
        assert got[0][0] == ref[0][0] and got[1] == ref[1] and got[3] == ref[3]
    # one batch per lock-step iteration, plus a follow-up batch when some run needs its expansion or shrink
    # point (these starts are far from their minima, so a few do): far fewer round trips than evaluations
    deepest = max(r[2] for r in together)
    assert deepest <= len(batches) <= deepest + 8
    assert len(batches) < sum(r[3] for r in together) / 3
    assert sum(r[3] for r in together) > 3 * len(batches)


def test_numpy_scalar_and_array_starts():
    f = _objectives()[4]
    a = fmin_lockstep(lambda xs: np.array([f(x) for x in xs]), [np.float64(169.2)])[0]
    b = fmin_lockstep(lambda xs: np.array([f(x) for x in xs]), [np.array([169.2])])[0]
    assert a[0][0] == b[0][0] and isinstance(a[0][0], np.float64)


def test_maxfun_termination_matches_scipy():
    f = _objectives()[2]
    ref = fmin(lambda x: float(f(x[0])), 0.4, full_output=True, disp=False, maxfun=11)
    got = fmin_lockstep(lambda xs: np.array([f(x) for x in xs]), [0.4], maxfun=11, maxiter=np.inf)[0]
    assert got[0][0] == ref[0][0] and got[1] == ref[1] and got[3] == ref[3]


@pytest.mark.parametrize("engine", ["NativeNelderMead", "NelderMeadCoreOnHost"])
def test_native_neldermead_equals_the_python_generator(engine):
    """csrc/parrm_nm.hip (the refinement inside the C library: ``parrm_nm_next`` / ``parrm_nm_feed``, and through them
    ``parrm_nm_minimise_fit``) against ``fmin_lockstep_requests``: the same batches of abscissae in the same order and
    the same (xopt, fopt, iterations, funcalls) on 400 random problems -- smooth, rough, NaN plateaus, runs that end on
    maxiter / maxfun, exact ties, a zero start, one to five starts, random tolerances.  Needs no GPU.

    ``NelderMeadCoreOnHost``: the same for ``csrc/parrm_nm_core.h``, the plain-data restatement of that state machine
    which the DEVICE runs between two optimiser batches (``nm_chain_step_kernel``), compiled for the host."""
    from pyparrm_amd import _hip
    from pyparrm_amd._neldermead import fmin_lockstep_requests

    rng = np.random.default_rng(0)

    def objective(kind, c):
        if kind == 0:
            return lambda x: (x - c) ** 2 * 1e-3 + 1.0 + 1e-4 * np.sin(40 * x)
        if kind == 1:
            return lambda x: np.abs(x - c) ** 1.5 + 0.3 * np.cos(7 * x)
        if kind == 2:
            return lambda x: np.where(np.abs(x - c) < 0.3, np.nan, (x - c) ** 2)  # a NaN plateau
        if kind == 3:
            return lambda x: -np.exp(-((x - c) ** 2)) + 1e-2 * x  # flat far out: runs into maxfun
        return lambda x: np.floor(5 * (x - c)) ** 2 + 0.0  # plateaus: exact ties

    batches = 0
    for case in range(400):
        c = rng.uniform(-5, 200)
        f = objective(case % 5, c)
        n = int(rng.integers(1, 6))
        starts = list(c + rng.uniform(-3, 3, n)) if case % 7 else [0.0] + list(c + rng.uniform(-1, 1, n - 1))
        kw = dict(xtol=10.0 ** rng.integers(-6, -2), ftol=10.0 ** rng.integers(-6, -2))
        if case % 11 == 0:
            kw.update(maxiter=int(rng.integers(1, 30)), maxfun=int(rng.integers(2, 40)))
        steps = fmin_lockstep_requests(starts, kw["xtol"], kw["ftol"], kw.get("maxiter"), kw.get("maxfun"))
        native = getattr(_hip, engine)(starts, **kw)
        try:
            points = next(steps)
            while True:
                mine = native.next_batch()
                assert mine is not None and np.array_equal(mine, points), (case, points, mine)
                values = np.asarray(f(points), dtype=np.float64)
                native.feed(values)
                batches += 1
                points = steps.send(values)
        except StopIteration as stop:
            expected = stop.value
        assert native.next_batch() is None, case
        for a, b in zip(expected, native.results()):
            assert np.array_equal(a[0], b[0], equal_nan=True), case
            assert a[1] == b[1] or (a[1] != a[1] and b[1] != b[1]), case
            assert a[2:] == b[2:], case
    assert batches > 3000
