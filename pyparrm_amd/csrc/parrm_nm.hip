// Lock-step one-dimensional Nelder-Mead with SciPy's semantics, native twin of pyparrm_amd/_neldermead.py.
//
// The reference refines period estimates with scipy.optimize.fmin (/root/reference/src/pyparrm/parrm.py:510-517,
// :545-550).  pyparrm_amd/_neldermead.py restates SciPy's run for one parameter (iterates, fopt, iteration and call
// counts equal to fmin's: tests/test_neldermead.py) as a generator of BATCHES of abscissae: the reflection and the two
// contractions of every run in flight (expansion / shrink points on demand), plus -- while at most two runs are in
// flight -- the points of the likely following step.  One batch is one device call (parrm_fit_errors_host), and with
// the device side of a batch at 50-170 us the ~12 us of Python between two batches (set building, sorting, the
// generator protocol, the ctypes call) were a tenth of the optimiser's time.  This file is that logic again, statement
// for statement -- same floating-point expressions in the same order, same table of known values, same batches in
// the same (ascending) order -- so that a whole search runs inside ONE C call (parrm_nm_minimise_fit); the step
// interface (parrm_nm_next / parrm_nm_feed) is what tests/test_neldermead.py drives against the Python generator.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <limits>
#include <unordered_map>
#include <vector>

#include "parrm_common.h"
#include "parrm_hip.h"
#include "parrm_nm_core.h"

namespace {

constexpr double kRho = 1, kChi = 2, kPsi = 0.5, kSigma = 0.5;  // SciPy's (non-adaptive) coefficients
constexpr double kNonZDelt = 0.05, kZDelt = 0.00025;

inline uint64_t key_of(double x) {
    if (x == 0.0) x = 0.0;  // -0.0 and 0.0 are one key, as in a Python dict
    uint64_t k;
    std::memcpy(&k, &x, sizeof k);
    return k;
}
inline bool is_nan(double x) { return x != x; }

struct Missing {
    double x;
};
struct MaxFun {};

using Table = std::unordered_map<uint64_t, double>;

// State of one SciPy run (_neldermead.py: _Start)
struct Run {
    double xatol, fatol;
    int maxiter, maxfun;
    double sim[2], fsim[2];
    int fcalls = 0, iterations = 0;
    bool done = false;
    double xr = 0, xe = 0, xc = 0, xcc = 0, xs = 0;

    Run(double x0, double xatol_, double fatol_, int maxiter_, int maxfun_)
        : xatol(xatol_), fatol(fatol_), maxiter(maxiter_ < 0 ? 200 : maxiter_), maxfun(maxfun_ < 0 ? 200 : maxfun_) {
        sim[0] = x0;
        sim[1] = x0 != 0 ? x0 * (1 + kNonZDelt) : kZDelt;
        fsim[0] = fsim[1] = std::numeric_limits<double>::infinity();
    }

    // (reflection, outside contraction, inside contraction) of the simplex {xbar, worst}
    static void trial_points(double xbar, double worst, double out[3]) {
        out[0] = (1 + kRho) * xbar - kRho * worst;
        out[1] = (1 + kPsi * kRho) * xbar - kPsi * kRho * worst;
        out[2] = (1 - kPsi) * xbar + kPsi * worst;
    }
    // abscissae of the step AFTER the next one, should the next one accept its inside contraction
    int lookahead(double out[6]) const {
        if (iterations == 0) return 0;
        const double best = sim[0], worst = sim[1];
        const double nw = (1 - kPsi) * best + kPsi * worst;
        trial_points(nw, best, out);
        trial_points(best, nw, out + 3);
        return 6;
    }
    // the abscissae the next step may need
    int wanted(double out[3]) {
        if (iterations == 0) {
            out[0] = sim[0];
            out[1] = sim[1];
            return 2;
        }
        const double xbar = sim[0], worst = sim[1];
        xr = (1 + kRho) * xbar - kRho * worst;
        xe = (1 + kRho * kChi) * xbar - kRho * kChi * worst;
        xc = (1 + kPsi * kRho) * xbar - kPsi * kRho * worst;
        xcc = (1 - kPsi) * xbar + kPsi * worst;
        xs = xbar + kSigma * (worst - xbar);
        out[0] = xr;
        out[1] = xc;
        out[2] = xcc;
        return 3;  // (expansion / shrink on demand: see advance())
    }
    double f(const Table &table, double x) {
        if (fcalls >= maxfun) throw MaxFun{};  // SciPy's wrapper refuses the call
        if (is_nan(x)) {
            ++fcalls;
            return std::numeric_limits<double>::quiet_NaN();
        }
        const auto it = table.find(key_of(x));
        if (it == table.end()) throw Missing{x};
        ++fcalls;
        return it->second;
    }
    void sort() {
        const double f0 = fsim[0], f1 = fsim[1];
        if (f1 < f0 || (is_nan(f0) && !is_nan(f1))) {  // argsort: ascending, stable, NaN last
            std::swap(sim[0], sim[1]);
            std::swap(fsim[0], fsim[1]);
        }
    }
    void check_done() {
        if (!(fcalls < maxfun && iterations < maxiter))
            done = true;
        else if (std::fabs(sim[1] - sim[0]) <= xatol && std::fabs(fsim[0] - fsim[1]) <= fatol)
            done = true;
    }
    // one SciPy step, reading the speculative results; throws Missing -- with this object unchanged -- when the step
    // needs an abscissa that is not in the table yet
    void advance(const Table &table) {
        const Run saved = *this;
        try {
            advance_inner(table);
        } catch (const Missing &) {
            const double keep[5] = {xr, xe, xc, xcc, xs};
            *this = saved;
            xr = keep[0], xe = keep[1], xc = keep[2], xcc = keep[3], xs = keep[4];
            throw;
        }
    }
    void advance_inner(const Table &table) {
        if (iterations == 0) {
            try {
                for (int k = 0; k < 2; ++k) fsim[k] = f(table, sim[k]);
            } catch (const MaxFun &) {
            }
            sort();
            iterations = 1;
        } else {
            try {
                const double fxr = f(table, xr);
                bool doshrink = false;
                if (fxr < fsim[0]) {
                    const double fxe = f(table, xe);
                    if (fxe < fxr) {
                        sim[1] = xe;
                        fsim[1] = fxe;
                    } else {
                        sim[1] = xr;
                        fsim[1] = fxr;
                    }
                } else {  // fsim[0] <= fxr; with one parameter fsim[-2] is fsim[0], so no plain accept
                    if (fxr < fsim[0]) {
                        sim[1] = xr;
                        fsim[1] = fxr;
                    } else {
                        if (fxr < fsim[1]) {
                            const double fxc = f(table, xc);
                            if (fxc <= fxr) {
                                sim[1] = xc;
                                fsim[1] = fxc;
                            } else {
                                doshrink = true;
                            }
                        } else {
                            const double fxcc = f(table, xcc);
                            if (fxcc < fsim[1]) {
                                sim[1] = xcc;
                                fsim[1] = fxcc;
                            } else {
                                doshrink = true;
                            }
                        }
                        if (doshrink) {
                            sim[1] = xs;
                            fsim[1] = f(table, xs);
                        }
                    }
                }
                ++iterations;
            } catch (const MaxFun &) {
            }
            sort();
        }
        check_done();
    }
    void result(double *xopt, double *fopt, int *its, int *calls) const {
        const double f0 = fsim[0], f1 = fsim[1];
        double fm = ((f0 <= f1 || is_nan(f1)) && !is_nan(f0)) ? f0 : (!is_nan(f1) ? f1 : std::numeric_limits<double>::quiet_NaN());
        if (is_nan(f0) || is_nan(f1)) fm = std::numeric_limits<double>::quiet_NaN();  // np.min propagates NaN
        *xopt = sim[0];
        *fopt = fm;
        *its = iterations;
        *calls = fcalls;
    }
};

}  // namespace

// the generator of _neldermead.py: fmin_lockstep_requests, as a state machine
struct parrm_nm {
    std::vector<Run> runs;
    Table table;
    int lookahead_runs = 2;
    // what the object was built from (the device-side chain starts its own copy of the state machine from these)
    std::vector<double> starts;
    double xatol = 0, fatol = 0;
    int maxiter = -1, maxfun = -1;
    bool fresh() const { return state == kTop && table.empty() && std::all_of(runs.begin(), runs.end(), [](const Run &r) {
                                    return r.iterations == 0 && r.fcalls == 0 && !r.done;
                                }); }
    // kTop: form the next main batch; kWait: a batch is out, its values are awaited; kAdvance: values are in, the
    // pending runs take their steps; kDone: every run has ended
    enum { kTop, kWait, kAdvance, kDone } state = kTop;
    std::vector<int> active, pending;
    std::vector<double> want, batch;

    static void sort_unique(std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end(), [](double a, double b) { return key_of(a) == key_of(b); }), v.end());
    }
    bool known(double x) const { return table.find(key_of(x)) != table.end(); }

    // Runs the generator up to its next `yield` (batch non-empty, state kWait) or to its end (batch empty, kDone).
    void produce() {
        batch.clear();
        while (true) {
            if (state == kDone) return;
            if (state == kTop) {
                active.clear();
                for (size_t i = 0; i < runs.size(); ++i)
                    if (!runs[i].done) active.push_back(static_cast<int>(i));
                if (active.empty()) {
                    state = kDone;
                    return;
                }
                want.clear();
                for (int i : active) {
                    double w[3];
                    const int n = runs[i].wanted(w);
                    for (int k = 0; k < n; ++k)
                        if (!is_nan(w[k])) want.push_back(w[k]);
                }
                sort_unique(want);
                std::vector<double> need;
                for (double x : want)
                    if (!known(x)) need.push_back(x);
                pending = active;
                state = kAdvance;
                if (!need.empty()) {
                    if (static_cast<int>(active.size()) <= lookahead_runs) {
                        for (int i : active) {
                            double a[6];
                            const int n = runs[i].lookahead(a);
                            for (int k = 0; k < n; ++k)
                                if (!is_nan(a[k]) && !known(a[k])) need.push_back(a[k]);
                        }
                        sort_unique(need);
                    }
                    batch = need;
                    state = kWait;
                    return;
                }
            }
            // kAdvance: the steps of the pending runs
            std::vector<double> missing;
            std::vector<int> still;
            for (int i : pending) {
                try {
                    runs[i].advance(table);
                } catch (const Missing &m) {
                    missing.push_back(m.x);
                    still.push_back(i);
                }
            }
            if (missing.empty()) {
                if (table.size() > 4096) {  // (a search never gets near this; keeps a pathological caller bounded)
                    Table kept;
                    for (double x : want) {
                        const auto it = table.find(key_of(x));
                        if (it != table.end()) kept.emplace(it->first, it->second);
                    }
                    table.swap(kept);
                }
                state = kTop;
                continue;
            }
            sort_unique(missing);  // expansion / shrink points of the few runs that need them
            batch = missing;
            pending = still;
            state = kWait;
            return;
        }
    }
    void feed(const double *values) {
        for (size_t i = 0; i < batch.size(); ++i) table[key_of(batch[i])] = values[i];
        state = kAdvance;
    }
};

namespace {
std::atomic<long long> g_chain_runs{0}, g_stepped_runs{0}, g_chain_batches{0}, g_stepped_batches{0};
}

extern "C" {

int parrm_nm_chain_stats(long long out[4]) {
    PARRM_REQUIRE(out, "nm_chain_stats: NULL argument");
    out[0] = g_chain_runs.load();
    out[1] = g_stepped_runs.load();
    out[2] = g_chain_batches.load();
    out[3] = g_stepped_batches.load();
    return PARRM_OK;
}

int parrm_nm_create(const double *starts, int n_starts, double xatol, double fatol, int maxiter, int maxfun,
                    int lookahead_runs, parrm_nm **nm) {
    try {
        PARRM_REQUIRE(starts && nm && n_starts > 0, "nm_create: NULL argument or no start");
        auto *p = new parrm_nm();
        p->lookahead_runs = lookahead_runs;
        p->starts.assign(starts, starts + n_starts);
        p->xatol = xatol, p->fatol = fatol, p->maxiter = maxiter, p->maxfun = maxfun;
        for (int i = 0; i < n_starts; ++i) p->runs.emplace_back(starts[i], xatol, fatol, maxiter, maxfun);
        *nm = p;
        return PARRM_OK;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary: std::bad_alloc from the table or the batch vectors)
        parrm::set_error("parrm_nm_create: %s", e.what());
        return PARRM_ERR_INTERNAL;
    }
}

int parrm_nm_destroy(parrm_nm *nm) {
    delete nm;
    return PARRM_OK;
}

int parrm_nm_next(parrm_nm *nm, double *points, int capacity, int *n) {
    try {
        PARRM_REQUIRE(nm && n, "nm_next: NULL argument");
        PARRM_REQUIRE(nm->state != parrm_nm::kWait, "nm_next: the last batch has not been fed");
        nm->produce();
        *n = static_cast<int>(nm->batch.size());
        PARRM_REQUIRE(*n <= capacity && (*n == 0 || points), "nm_next: %d points do not fit the buffer of %d", *n, capacity);
        if (*n) std::memcpy(points, nm->batch.data(), nm->batch.size() * sizeof(double));
        return PARRM_OK;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary: std::bad_alloc from the table or the batch vectors)
        parrm::set_error("parrm_nm_next: %s", e.what());
        return PARRM_ERR_INTERNAL;
    }
}

int parrm_nm_feed(parrm_nm *nm, const double *values, int n) {
    try {
        PARRM_REQUIRE(nm && values, "nm_feed: NULL argument");
        PARRM_REQUIRE(nm->state == parrm_nm::kWait && n == static_cast<int>(nm->batch.size()),
                      "nm_feed: no batch of %d points is waiting", n);
        nm->feed(values);
        return PARRM_OK;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary: std::bad_alloc from the table or the batch vectors)
        parrm::set_error("parrm_nm_feed: %s", e.what());
        return PARRM_ERR_INTERNAL;
    }
}

int parrm_nm_result(const parrm_nm *nm, int run, double *xopt, double *fopt, int *iterations, int *funcalls) {
    PARRM_REQUIRE(nm && xopt && fopt && iterations && funcalls, "nm_result: NULL argument");
    PARRM_REQUIRE(run >= 0 && run < static_cast<int>(nm->runs.size()), "nm_result: no run %d", run);
    nm->runs[run].result(xopt, fopt, iterations, funcalls);
    return PARRM_OK;
}

// Largest batch a search makes: 3 abscissae per run (<= 5 runs in the reference's stages) + 6 of look-ahead for
// each of <= 2 runs, or a follow-up batch of <= 1 per run -- 64 leaves room for other callers.
enum { kNmMaxBatch = 64 };

size_t parrm_nm_fit_workspace_bytes(int64_t n_idx, int64_t n_chans, int bw) {
    size_t most = parrm::nm_chain_workspace_bytes(n_idx, n_chans, bw);  // (0: a shape the device-side chain does not take)
    for (int n = 1; n <= kNmMaxBatch; ++n) {
        const size_t b = parrm_fit_workspace_bytes(n_idx, n_chans, n, bw);
        if (b == 0) return 0;
        most = std::max(most, b + 2 * static_cast<size_t>(n) * sizeof(double));
    }
    return most;
}

int parrm_nm_minimise_fit(parrm_nm *nm, const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                          int64_t n_chans, int bw, double lambda, void *d_workspace, size_t workspace_bytes, void *stream,
                          double *hist_x, double *hist_f, int hist_capacity, int *batch_sizes, int batch_capacity,
                          int *n_batches) {
    try {
        PARRM_REQUIRE(nm && d_y && d_idx && d_workspace && n_batches, "nm_minimise_fit: NULL argument");
        PARRM_REQUIRE(nm->state != parrm_nm::kWait, "nm_minimise_fit: a batch of this search is waiting for its values");
        // The refinement as a device-side chain: the device carries the state machine (parrm_nm_core.h) from batch to batch
        // and the host only keeps the queue filled.  The host does NOT take the result on trust: the recorded batches are
        // replayed through this object's own state machine (written independently: exceptions, a hash map) and the run is
        // refused unless every batch's abscissae and every run's result agree bit for bit.
        if (nm->fresh() && static_cast<int>(nm->runs.size()) <= parrm_nmcore::kMaxRuns) {
            static thread_local std::vector<double> cx, cf;
            static thread_local std::vector<int> csz;
            cx.resize(4096), cf.resize(4096), csz.resize(1024);
            parrm_nmcore::Core init;
            std::memset(&init, 0, sizeof init);
            parrm_nmcore::core_init(init, nm->starts.data(), static_cast<int>(nm->starts.size()), nm->xatol, nm->fatol, nm->maxiter,
                                    nm->maxfun, nm->lookahead_runs);
            double rx[parrm_nmcore::kMaxRuns], rf[parrm_nmcore::kMaxRuns];
            int rits[parrm_nmcore::kMaxRuns], rcalls[parrm_nmcore::kMaxRuns];
            int nb = 0, used = 0;
            bool handled = false;
            const int rc = parrm::nm_chain_run(init, d_y, ldy, d_idx, n_idx, n_chans, bw, lambda, d_workspace, workspace_bytes, stream,
                                               cx.data(), cf.data(), 4096, csz.data(), 1024, &nb, &used, rx, rf, rits, rcalls, &handled);
            if (rc != PARRM_OK) return rc;
            if (handled) {
                int at = 0;
                for (int b = 0; b < nb; ++b) {
                    nm->produce();
                    const int n = static_cast<int>(nm->batch.size());
                    bool same = n == csz[b] && n > 0;
                    for (int i = 0; same && i < n; ++i) same = key_of(nm->batch[i]) == key_of(cx[at + i]);
                    if (!same) {
                        parrm::set_error("nm_minimise_fit: batch %d of the device-side refinement differs from the host's replay", b);
                        return PARRM_ERR_INTERNAL;
                    }
                    nm->feed(cf.data() + at);
                    at += n;
                }
                nm->produce();
                bool same = nm->batch.empty() && nm->state == parrm_nm::kDone && at == used;
                for (size_t r = 0; same && r < nm->runs.size(); ++r) {
                    double x, f;
                    int its, calls;
                    nm->runs[r].result(&x, &f, &its, &calls);
                    same = key_of(x) == key_of(rx[r]) && (key_of(f) == key_of(rf[r]) || (is_nan(f) && is_nan(rf[r]))) && its == rits[r] &&
                           calls == rcalls[r];
                }
                if (!same) {
                    parrm::set_error("nm_minimise_fit: the device-side refinement's results differ from the host's replay");
                    return PARRM_ERR_INTERNAL;
                }
                if (hist_x && hist_f && batch_sizes && used <= hist_capacity && nb <= batch_capacity) {
                    std::memcpy(hist_x, cx.data(), static_cast<size_t>(used) * sizeof(double));
                    std::memcpy(hist_f, cf.data(), static_cast<size_t>(used) * sizeof(double));
                    std::memcpy(batch_sizes, csz.data(), static_cast<size_t>(nb) * sizeof(int));
                }
                *n_batches = nb;
                ++g_chain_runs;
                g_chain_batches += nb;
                PARRM_REQUIRE(!hist_x || (used <= hist_capacity && nb <= batch_capacity),
                              "nm_minimise_fit: the history (%d evaluations in %d batches) does not fit the buffers", used, nb);
                return PARRM_OK;
            }
        }
        double x[kNmMaxBatch], f[kNmMaxBatch];
        int used = 0, batches = 0;
        while (true) {
            nm->produce();
            const int n = static_cast<int>(nm->batch.size());
            if (n == 0) break;
            PARRM_REQUIRE(n <= kNmMaxBatch, "nm_minimise_fit: a batch of %d abscissae (more than %d)", n, kNmMaxBatch);
            std::memcpy(x, nm->batch.data(), n * sizeof(double));
            const int rc = parrm_fit_errors_host(d_y, ldy, d_idx, n_idx, n_chans, x, n, bw, lambda, f, d_workspace, workspace_bytes,
                                                 stream);
            if (rc != PARRM_OK) return rc;
            nm->feed(f);
            if (hist_x && hist_f && batch_sizes && used + n <= hist_capacity && batches < batch_capacity) {
                std::memcpy(hist_x + used, x, n * sizeof(double));
                std::memcpy(hist_f + used, f, n * sizeof(double));
                batch_sizes[batches] = n;
            }
            used += n;
            ++batches;
        }
        *n_batches = batches;
        ++g_stepped_runs;
        g_stepped_batches += batches;
        PARRM_REQUIRE(!hist_x || (used <= hist_capacity && batches <= batch_capacity),
                      "nm_minimise_fit: the history (%d evaluations in %d batches) does not fit the buffers", used, batches);
        return PARRM_OK;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary: std::bad_alloc from the table or the batch vectors)
        parrm::set_error("parrm_nm_minimise_fit: %s", e.what());
        return PARRM_ERR_INTERNAL;
    }
}

// ---- the plain-data state machine the DEVICE runs (parrm_nm_core.h), compiled for the host: test surface ------------
// tests/test_neldermead.py drives these against the Python generator exactly as it drives parrm_nm_* above; no GPU.
struct parrm_nmcore_handle {
    parrm_nmcore::Core core;
    std::vector<uint64_t> keys;
    std::vector<double> vals;
    int n = 0;
    parrm_nmcore::FlatTable table() { return parrm_nmcore::FlatTable{keys.data(), vals.data(), &n, static_cast<int>(keys.size())}; }
};

int parrm_nmcore_create(const double *starts, int n_starts, double xatol, double fatol, int maxiter, int maxfun,
                        int lookahead_runs, void **handle) {
    try {
        PARRM_REQUIRE(starts && handle && n_starts > 0 && n_starts <= parrm_nmcore::kMaxRuns, "nmcore_create: 1..%d starts",
                      parrm_nmcore::kMaxRuns);
        auto *h = new parrm_nmcore_handle();
        std::memset(&h->core, 0, sizeof h->core);
        parrm_nmcore::core_init(h->core, starts, n_starts, xatol, fatol, maxiter, maxfun, lookahead_runs);
        h->keys.resize(1 << 16);
        h->vals.resize(1 << 16);
        *handle = h;
        return PARRM_OK;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary: std::bad_alloc from the table or the batch vectors)
        parrm::set_error("parrm_nmcore_create: %s", e.what());
        return PARRM_ERR_INTERNAL;
    }
}

int parrm_nmcore_destroy(void *handle) {
    delete static_cast<parrm_nmcore_handle *>(handle);
    return PARRM_OK;
}

int parrm_nmcore_next(void *handle, double *points, int capacity, int *n) {
    auto *h = static_cast<parrm_nmcore_handle *>(handle);
    PARRM_REQUIRE(h && n, "nmcore_next: NULL argument");
    PARRM_REQUIRE(h->core.state != parrm_nmcore::kWait, "nmcore_next: the last batch has not been fed");
    auto t = h->table();
    parrm_nmcore::core_produce(h->core, t);
    PARRM_REQUIRE(h->core.error == 0, "nmcore_next: the state machine gave up (code %d)", h->core.error);
    *n = h->core.n_batch;
    PARRM_REQUIRE(*n <= capacity && (*n == 0 || points), "nmcore_next: %d points do not fit the buffer of %d", *n, capacity);
    if (*n) std::memcpy(points, h->core.batch, static_cast<size_t>(*n) * sizeof(double));
    return PARRM_OK;
}

int parrm_nmcore_feed(void *handle, const double *values, int n) {
    auto *h = static_cast<parrm_nmcore_handle *>(handle);
    PARRM_REQUIRE(h && values, "nmcore_feed: NULL argument");
    PARRM_REQUIRE(h->core.state == parrm_nmcore::kWait && n == h->core.n_batch, "nmcore_feed: no batch of %d points is waiting", n);
    auto t = h->table();
    parrm_nmcore::core_feed(h->core, t, values);
    PARRM_REQUIRE(h->core.error == 0, "nmcore_feed: the table is full");
    return PARRM_OK;
}

int parrm_nmcore_result(void *handle, int run, double *xopt, double *fopt, int *iterations, int *funcalls) {
    auto *h = static_cast<parrm_nmcore_handle *>(handle);
    PARRM_REQUIRE(h && xopt && fopt && iterations && funcalls, "nmcore_result: NULL argument");
    PARRM_REQUIRE(run >= 0 && run < h->core.n_runs, "nmcore_result: no run %d", run);
    parrm_nmcore::run_result(h->core.runs[run], xopt, fopt, iterations, funcalls);
    return PARRM_OK;
}

}  // extern "C"
