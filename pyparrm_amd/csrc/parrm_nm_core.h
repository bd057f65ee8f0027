// The lock-step one-dimensional Nelder-Mead of parrm_nm.hip / pyparrm_amd/_neldermead.py once more, as plain data and
// plain functions that compile for the HOST and for the DEVICE: no exceptions, no containers, no allocation.
//
// Why a third statement of the same rule: since round 4 the refinement's state lives on the device
// (parrm_period.hip: nm_chain_step_kernel) -- the kernel that closes an optimiser batch also takes SciPy's decisions
// and leaves the next batch's abscissae in device memory, so the host is out of the loop between two batches
// (/root/reference/src/pyparrm/parrm.py:499-517, :545-550 is a chain of ~76 dependent batches).  The code below is what
// that kernel runs.  parrm_nm.hip (exceptions, std::unordered_map) stays the independent host twin: after a device
// run the host REPLAYS the recorded batches through it and refuses the result unless every batch and every decision
// agree; tests/test_neldermead.py drives this core (compiled for the host, parrm_nmcore_* entry points) against the
// Python generator on the same 400 random problems as the twin.
//
// Same floating-point expressions in the same order as parrm_nm.hip (the library is built with -ffp-contract=off for
// host and device code alike), same table semantics (keys are the doubles' bits, -0.0 folded onto 0.0), same batches
// in the same ascending order.
#pragma once

#include <cstdint>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define PARRM_HD __host__ __device__ inline
#else
#define PARRM_HD inline
#endif

namespace parrm_nmcore {

constexpr int kMaxRuns = 8;    // runs of one lock-step refinement (the reference starts <= 5, parrm.py:499)
constexpr int kMaxBatch = 32;  // abscissae per batch: 3 per run + 6 of look-ahead for each of <= 2 runs
constexpr double kRho = 1, kChi = 2, kPsi = 0.5, kSigma = 0.5;  // SciPy's (non-adaptive) coefficients
constexpr double kNonZDelt = 0.05, kZDelt = 0.00025;

enum { kTop = 0, kWait = 1, kAdvance = 2, kDone = 3 };
enum { kErrNone = 0, kErrBatchTooLarge = 1, kErrTableFull = 2 };

PARRM_HD uint64_t key_of(double x) {
    if (x == 0.0) x = 0.0;  // -0.0 and 0.0 are one key, as in a Python dict
    union {
        double d;
        uint64_t u;
    } c;
    c.d = x;
    return c.u;
}
PARRM_HD bool is_nan(double x) { return x != x; }
PARRM_HD double nan_value() {
    union {
        uint64_t u;
        double d;
    } c;
    c.u = 0x7ff8000000000000ull;
    return c.d;
}
PARRM_HD double inf_value() {
    union {
        uint64_t u;
        double d;
    } c;
    c.u = 0x7ff0000000000000ull;
    return c.d;
}
PARRM_HD double abs_value(double x) { return x < 0 ? -x : (x == 0 ? 0.0 : x); }  // (NaN never reaches it)

// State of one SciPy run (_neldermead.py: _Start; parrm_nm.hip: Run)
struct Run {
    double xatol, fatol;
    double sim[2], fsim[2];
    double xr, xe, xc, xcc, xs;
    int maxiter, maxfun;
    int fcalls, iterations, done, pad_;
};

struct Core {
    Run runs[kMaxRuns];
    double batch[kMaxBatch];  // the batch that is out (state kWait), ascending
    int n_runs, lookahead_runs, state, n_batch;
    unsigned pending;  // runs whose step is due (bit i = run i; steps are taken in index order)
    int error, pad_;
};

PARRM_HD void run_init(Run &r, double x0, double xatol, double fatol, int maxiter, int maxfun) {
    r.xatol = xatol;
    r.fatol = fatol;
    r.maxiter = maxiter < 0 ? 200 : maxiter;
    r.maxfun = maxfun < 0 ? 200 : maxfun;
    r.sim[0] = x0;
    r.sim[1] = x0 != 0 ? x0 * (1 + kNonZDelt) : kZDelt;
    r.fsim[0] = r.fsim[1] = inf_value();
    r.fcalls = r.iterations = r.done = r.pad_ = 0;
    r.xr = r.xe = r.xc = r.xcc = r.xs = 0;
}

PARRM_HD void core_init(Core &c, const double *starts, int n_starts, double xatol, double fatol, int maxiter, int maxfun,
                        int lookahead_runs) {
    c.n_runs = n_starts;
    c.lookahead_runs = lookahead_runs;
    c.state = kTop;
    c.n_batch = 0;
    c.pending = 0;
    c.error = kErrNone;
    c.pad_ = 0;
    for (int i = 0; i < n_starts; ++i) run_init(c.runs[i], starts[i], xatol, fatol, maxiter, maxfun);
}

// (reflection, outside contraction, inside contraction) of the simplex {xbar, worst}
PARRM_HD void trial_points(double xbar, double worst, double out[3]) {
    out[0] = (1 + kRho) * xbar - kRho * worst;
    out[1] = (1 + kPsi * kRho) * xbar - kPsi * kRho * worst;
    out[2] = (1 - kPsi) * xbar + kPsi * worst;
}
// abscissae of the step AFTER the next one, should the next one accept its inside contraction
PARRM_HD int run_lookahead(const Run &r, double out[6]) {
    if (r.iterations == 0) return 0;
    const double best = r.sim[0], worst = r.sim[1];
    const double nw = (1 - kPsi) * best + kPsi * worst;
    trial_points(nw, best, out);
    trial_points(best, nw, out + 3);
    return 6;
}
// the abscissae the next step may need
PARRM_HD int run_wanted(Run &r, double out[3]) {
    if (r.iterations == 0) {
        out[0] = r.sim[0];
        out[1] = r.sim[1];
        return 2;
    }
    const double xbar = r.sim[0], worst = r.sim[1];
    r.xr = (1 + kRho) * xbar - kRho * worst;
    r.xe = (1 + kRho * kChi) * xbar - kRho * kChi * worst;
    r.xc = (1 + kPsi * kRho) * xbar - kPsi * kRho * worst;
    r.xcc = (1 - kPsi) * xbar + kPsi * worst;
    r.xs = xbar + kSigma * (worst - xbar);
    out[0] = r.xr;
    out[1] = r.xc;
    out[2] = r.xcc;
    return 3;  // (expansion / shrink on demand: see run_advance)
}
PARRM_HD void run_sort(Run &r) {
    const double f0 = r.fsim[0], f1 = r.fsim[1];
    if (f1 < f0 || (is_nan(f0) && !is_nan(f1))) {  // argsort: ascending, stable, NaN last
        const double s = r.sim[0], f = r.fsim[0];
        r.sim[0] = r.sim[1];
        r.fsim[0] = r.fsim[1];
        r.sim[1] = s;
        r.fsim[1] = f;
    }
}
PARRM_HD void run_check_done(Run &r) {
    if (!(r.fcalls < r.maxfun && r.iterations < r.maxiter))
        r.done = 1;
    else if (abs_value(r.sim[1] - r.sim[0]) <= r.xatol && abs_value(r.fsim[0] - r.fsim[1]) <= r.fatol)
        r.done = 1;
}

// One objective value: 0 = *val set, 1 = not in the table (*missing set), 2 = SciPy's wrapper refuses the call (maxfun)
template <class Table>
PARRM_HD int run_f(Run &r, const Table &table, double x, double *val, double *missing) {
    if (r.fcalls >= r.maxfun) return 2;
    if (is_nan(x)) {
        ++r.fcalls;
        *val = nan_value();
        return 0;
    }
    if (!table.find(x, val)) {
        *missing = x;
        return 1;
    }
    ++r.fcalls;
    return 0;
}

// One SciPy step reading the speculative results: 0 = taken, 1 = needs an abscissa that is not in the table yet
// (*missing; the run is left as it was).  (parrm_nm.hip: Run::advance / advance_inner.)
template <class Table>
PARRM_HD int run_advance(Run &r, const Table &table, double *missing) {
    const Run saved = r;
    int s = 0;
    double v = 0;
    bool missed = false;
    if (r.iterations == 0) {
        for (int k = 0; k < 2 && !missed; ++k) {
            s = run_f(r, table, r.sim[k], &v, missing);
            if (s == 1) missed = true;
            if (s != 0) break;  // (2: the calls that were made stand)
            r.fsim[k] = v;
        }
        if (!missed) {
            run_sort(r);
            r.iterations = 1;
        }
    } else {
        do {  // (one pass; `break` = SciPy's wrapper refused a call: the step ends where it stands)
            double fxr = 0;
            s = run_f(r, table, r.xr, &fxr, missing);
            if (s == 1) missed = true;
            if (s != 0) break;
            bool doshrink = false;
            if (fxr < r.fsim[0]) {
                double fxe = 0;
                s = run_f(r, table, r.xe, &fxe, missing);
                if (s == 1) missed = true;
                if (s != 0) break;
                if (fxe < fxr) {
                    r.sim[1] = r.xe;
                    r.fsim[1] = fxe;
                } else {
                    r.sim[1] = r.xr;
                    r.fsim[1] = fxr;
                }
            } else {  // fsim[0] <= fxr; with one parameter fsim[-2] is fsim[0], so no plain accept
                if (fxr < r.fsim[1]) {
                    double fxc = 0;
                    s = run_f(r, table, r.xc, &fxc, missing);
                    if (s == 1) missed = true;
                    if (s != 0) break;
                    if (fxc <= fxr) {
                        r.sim[1] = r.xc;
                        r.fsim[1] = fxc;
                    } else {
                        doshrink = true;
                    }
                } else {
                    double fxcc = 0;
                    s = run_f(r, table, r.xcc, &fxcc, missing);
                    if (s == 1) missed = true;
                    if (s != 0) break;
                    if (fxcc < r.fsim[1]) {
                        r.sim[1] = r.xcc;
                        r.fsim[1] = fxcc;
                    } else {
                        doshrink = true;
                    }
                }
                if (doshrink) {
                    r.sim[1] = r.xs;
                    s = run_f(r, table, r.xs, &v, missing);
                    if (s == 1) missed = true;
                    if (s != 0) break;
                    r.fsim[1] = v;
                }
            }
            ++r.iterations;
        } while (false);
        if (!missed) run_sort(r);
    }
    if (missed) {
        const double keep[5] = {r.xr, r.xe, r.xc, r.xcc, r.xs};
        r = saved;
        r.xr = keep[0], r.xe = keep[1], r.xc = keep[2], r.xcc = keep[3], r.xs = keep[4];
        return 1;
    }
    run_check_done(r);
    return 0;
}

PARRM_HD void run_result(const Run &r, double *xopt, double *fopt, int *its, int *calls) {
    const double f0 = r.fsim[0], f1 = r.fsim[1];
    double fm = ((f0 <= f1 || is_nan(f1)) && !is_nan(f0)) ? f0 : (!is_nan(f1) ? f1 : nan_value());
    if (is_nan(f0) || is_nan(f1)) fm = nan_value();  // np.min propagates NaN
    *xopt = r.sim[0];
    *fopt = fm;
    *its = r.iterations;
    *calls = r.fcalls;
}

// ascending, one entry per key (insertion sort: <= kMaxBatch + a few entries)
PARRM_HD int sort_unique(double *v, int n) {
    for (int i = 1; i < n; ++i) {
        const double x = v[i];
        int j = i - 1;
        while (j >= 0 && x < v[j]) {
            v[j + 1] = v[j];
            --j;
        }
        v[j + 1] = x;
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (m == 0 || key_of(v[i]) != key_of(v[m - 1])) v[m++] = v[i];
    return m;
}

// The generator of _neldermead.py (fmin_lockstep_requests) as a state machine: runs up to its next `yield`
// (c.n_batch > 0, state kWait) or to its end (c.n_batch == 0, state kDone).  (parrm_nm.hip: parrm_nm::produce.)
template <class Table>
PARRM_HD void core_produce(Core &c, const Table &table) {
    c.n_batch = 0;
    while (true) {
        if (c.state == kDone) return;
        if (c.state == kTop) {
            unsigned active = 0;
            int n_active = 0;
            for (int i = 0; i < c.n_runs; ++i)
                if (!c.runs[i].done) {
                    active |= 1u << i;
                    ++n_active;
                }
            if (n_active == 0) {
                c.state = kDone;
                return;
            }
            double want[3 * kMaxRuns];
            int n_want = 0;
            for (int i = 0; i < c.n_runs; ++i) {
                if (!(active >> i & 1u)) continue;
                double w[3];
                const int n = run_wanted(c.runs[i], w);
                for (int k = 0; k < n; ++k)
                    if (!is_nan(w[k])) want[n_want++] = w[k];
            }
            n_want = sort_unique(want, n_want);
            double need[3 * kMaxRuns + 6 * kMaxRuns];
            int n_need = 0;
            for (int k = 0; k < n_want; ++k)
                if (!table.known(want[k])) need[n_need++] = want[k];
            c.pending = active;
            c.state = kAdvance;
            if (n_need > 0) {
                if (n_active <= c.lookahead_runs) {
                    for (int i = 0; i < c.n_runs; ++i) {
                        if (!(active >> i & 1u)) continue;
                        double a[6];
                        const int n = run_lookahead(c.runs[i], a);
                        for (int k = 0; k < n; ++k)
                            if (!is_nan(a[k]) && !table.known(a[k])) need[n_need++] = a[k];
                    }
                    n_need = sort_unique(need, n_need);
                }
                if (n_need > kMaxBatch) {
                    c.error = kErrBatchTooLarge;
                    c.state = kDone;
                    return;
                }
                for (int k = 0; k < n_need; ++k) c.batch[k] = need[k];
                c.n_batch = n_need;
                c.state = kWait;
                return;
            }
        }
        // kAdvance: the steps of the pending runs
        double missing[kMaxRuns];
        int n_missing = 0;
        unsigned still = 0;
        for (int i = 0; i < c.n_runs; ++i) {
            if (!(c.pending >> i & 1u)) continue;
            double m = 0;
            if (run_advance(c.runs[i], table, &m)) {
                missing[n_missing++] = m;
                still |= 1u << i;
            }
        }
        if (n_missing == 0) {
            c.state = kTop;
            continue;
        }
        n_missing = sort_unique(missing, n_missing);  // expansion / shrink points of the few runs that need them
        for (int k = 0; k < n_missing; ++k) c.batch[k] = missing[k];
        c.n_batch = n_missing;
        c.pending = still;
        c.state = kWait;
        return;
    }
}

// values of the batch that is out -> table; the pending runs take their steps at the next core_produce
template <class Table>
PARRM_HD void core_feed(Core &c, Table &table, const double *values) {
    for (int i = 0; i < c.n_batch; ++i)
        if (!table.insert(c.batch[i], values[i])) c.error = kErrTableFull;
    c.state = kAdvance;
}

// A table over two flat arrays (host form: linear search; the device form in parrm_period.hip searches with the
// whole wave).  An abscissa fed twice overwrites its value, as a dict does.
struct FlatTable {
    uint64_t *keys;
    double *vals;
    int *n;
    int capacity;
    PARRM_HD int index_of(double x) const {
        const uint64_t k = key_of(x);
        for (int i = 0; i < *n; ++i)
            if (keys[i] == k) return i;
        return -1;
    }
    PARRM_HD bool known(double x) const { return index_of(x) >= 0; }
    PARRM_HD bool find(double x, double *val) const {
        const int i = index_of(x);
        if (i < 0) return false;
        *val = vals[i];
        return true;
    }
    PARRM_HD bool insert(double x, double val) {
        const int i = index_of(x);
        if (i >= 0) {
            vals[i] = val;
            return true;
        }
        if (*n >= capacity) return false;
        keys[*n] = key_of(x);
        vals[*n] = val;
        ++*n;
        return true;
    }
};

}  // namespace parrm_nmcore
