"""One-dimensional Nelder-Mead with SciPy's exact semantics, restructured for a GPU objective.

The reference refines period estimates with ``scipy.optimize.fmin`` (parrm.py:510-517, :545-550):
~35 strictly sequential objective evaluations per start, five starts per stage.  On the device one
evaluation is latency-bound (three small launches + one read-back), so the sequential count -- not
the arithmetic -- sets the time.  This module keeps SciPy's decisions bit-for-bit
(``scipy/optimize/_optimize.py::_minimize_neldermead``, N = 1, adaptive=False: rho 1, chi 2,
psi 0.5, sigma 0.5, initial simplex {x0, 1.05*x0}, xatol = fatol = 1e-4, maxiter = maxfun = 200)
but changes HOW the objective is called:

* speculation: in 1-D an iteration evaluates the reflection and then at most one of
  {expansion, outside contraction, inside contraction} and possibly the shrink point.  All five
  abscissae are known at the start of the iteration; the three that SciPy's run almost always needs --
  reflection and the two contractions (measured on PARRM objectives: reflection 85/85 iterations, inside
  contraction 75, outside contraction 10, expansion 0, shrink 0) -- are evaluated as ONE batch and the
  iteration's branches then only read the results they would have computed; an iteration that turns
  out to need the expansion or the shrink point gets it from a small follow-up batch.  (Round 1
  evaluated all five: 25-40 % more candidates per batch, which at 41 harmonics x 256 channels is real
  matrix-core time, not just latency.);
* lock-step starts: independent starts advance one iteration per batch, so a stage costs
  max(iterations) batches instead of sum(evaluations) launches.

Function-call accounting follows SciPy (only the evaluations SciPy would have made are counted
against ``maxfun``), so termination is identical.  tests/test_neldermead.py checks equality of
(xopt, fopt, iterations, funcalls) with ``scipy.optimize.fmin`` on deterministic objectives.
"""

from __future__ import annotations

import numpy as np

RHO, CHI, PSI, SIGMA = 1, 2, 0.5, 0.5
NONZDELT, ZDELT = 0.05, 0.00025


class _MaxFun(Exception):
    pass


class _Missing(Exception):
    """The step needs an abscissa that was not in the speculative batch."""

    def __init__(self, x):
        super().__init__(x)
        self.x = x


class _Start:
    """State of one SciPy Nelder-Mead run on a scalar parameter.

    The two vertices and their values are kept as Python floats (IEEE doubles, the arithmetic
    NumPy's float64 scalars use): a lock-step batch advances up to five of these objects, and the
    dozen NumPy calls per object and iteration that 2-element arrays cost (~15 us) were most of the
    host time between two device batches.  Ordering follows ``np.argsort`` on two elements: stable,
    NaN last."""

    def __init__(self, x0, xatol, fatol, maxiter, maxfun):
        x0 = np.atleast_1d(x0).flatten()
        if not np.issubdtype(x0.dtype, np.inexact):
            x0 = x0.astype(np.float64)
        if x0.shape[0] != 1:
            raise ValueError("only scalar parameters are supported")
        if x0.dtype != np.float64:
            raise TypeError("the batched Nelder-Mead works in float64 (got a %s start)" % x0.dtype)
        self.xatol, self.fatol = xatol, fatol
        self.maxiter = 200 if maxiter is None else maxiter
        self.maxfun = 200 if maxfun is None else maxfun
        first = float(x0[0])
        second = first * (1 + NONZDELT) if first != 0 else ZDELT
        self.sim = [first, second]
        self.fsim = [np.inf, np.inf]
        self.fcalls = 0
        self.iterations = 0  # 0: initial simplex not evaluated yet
        self.done = False

    @staticmethod
    def _trial_points(xbar, worst):
        """(reflection, outside contraction, inside contraction) of the simplex {xbar, worst}."""
        return ((1 + RHO) * xbar - RHO * worst,
                (1 + PSI * RHO) * xbar - PSI * RHO * worst,
                (1 - PSI) * xbar + PSI * worst)

    def lookahead(self):
        """Abscissae of the step AFTER the next one, should the next one accept its inside contraction
        -- what it does in ~9 of 10 steps -- with the new vertex ending up as either the better or the
        worse of the two.  A caller with few runs in flight adds them to the batch: when the guess holds
        the following step finds its values already there and costs no device round trip."""
        if self.iterations == 0:
            return []
        best, worst = self.sim
        new = (1 - PSI) * best + PSI * worst
        return list(self._trial_points(new, best)) + list(self._trial_points(best, new))

    # -- the abscissae the next step may need -------------------------------------------------
    def wanted(self):
        if self.iterations == 0:
            return list(self.sim)
        xbar, worst = self.sim
        self._xr = (1 + RHO) * xbar - RHO * worst
        self._xe = (1 + RHO * CHI) * xbar - RHO * CHI * worst
        self._xc = (1 + PSI * RHO) * xbar - PSI * RHO * worst
        self._xcc = (1 - PSI) * xbar + PSI * worst
        self._xs = xbar + SIGMA * (worst - xbar)
        return [self._xr, self._xc, self._xcc]  # (expansion / shrink on demand: see advance())

    def _f(self, table, x):
        if self.fcalls >= self.maxfun:  # SciPy's wrapper refuses the call
            raise _MaxFun()
        if x != x:
            self.fcalls += 1
            return float("nan")
        if x not in table:
            raise _Missing(x)
        self.fcalls += 1
        return table[x]

    def _sort(self):
        f0, f1 = self.fsim
        if f1 < f0 or (f0 != f0 and f1 == f1):  # argsort: ascending, stable, NaN last
            self.sim.reverse()
            self.fsim.reverse()

    # -- one SciPy step, reading the speculative results --------------------------------------
    def advance(self, table):
        """Raises ``_Missing`` -- with this object unchanged -- when the step needs an abscissa that is
        not in ``table`` yet; call again once it is."""
        saved = (list(self.sim), list(self.fsim), self.fcalls, self.iterations)
        try:
            self._advance(table)
        except _Missing:
            self.sim, self.fsim, self.fcalls, self.iterations = saved
            raise

    def _advance(self, table):
        if self.iterations == 0:
            try:
                for k in range(2):
                    self.fsim[k] = self._f(table, self.sim[k])
            except _MaxFun:
                pass
            self._sort()
            self.iterations = 1
        else:
            sim, fsim = self.sim, self.fsim
            try:
                fxr = self._f(table, self._xr)
                doshrink = False
                if fxr < fsim[0]:
                    fxe = self._f(table, self._xe)
                    if fxe < fxr:
                        sim[1], fsim[1] = self._xe, fxe
                    else:
                        sim[1], fsim[1] = self._xr, fxr
                else:  # fsim[0] <= fxr; with one parameter fsim[-2] is fsim[0], so no plain accept
                    if fxr < fsim[0]:
                        sim[1], fsim[1] = self._xr, fxr
                    else:
                        if fxr < fsim[1]:
                            fxc = self._f(table, self._xc)
                            if fxc <= fxr:
                                sim[1], fsim[1] = self._xc, fxc
                            else:
                                doshrink = True
                        else:
                            fxcc = self._f(table, self._xcc)
                            if fxcc < fsim[1]:
                                sim[1], fsim[1] = self._xcc, fxcc
                            else:
                                doshrink = True
                        if doshrink:
                            sim[1] = self._xs
                            fsim[1] = self._f(table, self._xs)
                self.iterations += 1
            except _MaxFun:
                pass
            self._sort()
        self._check_done()

    def _check_done(self):
        if not (self.fcalls < self.maxfun and self.iterations < self.maxiter):
            self.done = True
        elif abs(self.sim[1] - self.sim[0]) <= self.xatol and abs(self.fsim[0] - self.fsim[1]) <= self.fatol:
            self.done = True

    def result(self):
        f0, f1 = self.fsim
        fmin_ = f0 if (f0 <= f1 or f1 != f1) and f0 == f0 else (f1 if f1 == f1 else float("nan"))
        if f0 != f0 or f1 != f1:  # np.min propagates NaN
            fmin_ = float("nan")
        return np.array([self.sim[0]], dtype=np.float64), float(fmin_), self.iterations, self.fcalls


LOOKAHEAD_RUNS = 2  # at most this many runs in flight: also evaluate the likely next step's points


def fmin_lockstep_requests(starts, xtol=1e-4, ftol=1e-4, maxiter=None, maxfun=None):
    """Generator form of :func:`fmin_lockstep`: yields the float64 array of abscissae of each batch
    and expects their objective values to be sent back; returns (``StopIteration.value``) the list of
    ``(xopt[1], fopt, iterations, funcalls)``.  Lets a caller advance several independent searches
    together and evaluate all their batches in one device call.

    Values stay in a table across steps, and while only one or two runs are in flight (the final polish
    of a period search is a single run of ~35 dependent steps) each batch also carries the points of the
    likely FOLLOWING step (``_Start.lookahead``): a step whose points are all known already costs no
    round trip.  Which values a step reads, and so every decision, is unchanged."""
    runs = [_Start(x0, xtol, ftol, maxiter, maxfun) for x0 in starts]
    table = {}
    while True:
        active = [run for run in runs if not run.done]
        if not active:
            break
        want = {x for run in active for x in run.wanted() if x == x}
        need = sorted(x for x in want if x not in table)
        if need:
            if len(active) <= LOOKAHEAD_RUNS:
                ahead = {x for run in active for x in run.lookahead() if x == x and x not in table}
                need = sorted(set(need) | ahead)
            values = yield np.array(need, dtype=np.float64)
            table.update(zip(need, np.asarray(values, dtype=np.float64).tolist()))
        pending = active
        while pending:
            missing, still = set(), []
            for run in pending:
                try:
                    run.advance(table)
                except _Missing as miss:
                    missing.add(miss.x)
                    still.append(run)
            if not missing:
                break
            extra = sorted(missing)  # expansion / shrink points of the few runs that need them
            values = yield np.array(extra, dtype=np.float64)
            table.update(zip(extra, np.asarray(values, dtype=np.float64).tolist()))
            pending = still
        if len(table) > 4096:  # (a search never gets near this; keeps a pathological caller bounded)
            table = {x: table[x] for x in want if x in table}
    return [run.result() for run in runs]


def fmin_lockstep(objective_batch, starts, xtol=1e-4, ftol=1e-4, maxiter=None, maxfun=None):
    """Run one SciPy-equivalent ``fmin`` per entry of ``starts``, advancing all of them together.

    ``objective_batch(x: float64[n]) -> float64[n]`` evaluates the objective at every abscissa of
    a batch.  Returns a list of ``(xopt[1], fopt, iterations, funcalls)`` in the order of
    ``starts`` -- what ``fmin(..., full_output=True)[:4]`` returns.
    """
    requests = fmin_lockstep_requests(starts, xtol, ftol, maxiter, maxfun)
    try:
        points = next(requests)
        while True:
            points = requests.send(objective_batch(points))
    except StopIteration as stop:
        return stop.value
