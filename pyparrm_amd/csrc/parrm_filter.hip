// filter_data on gfx950: the phase-neighbourhood stencil of parrm.py:861-869 in closed form,
//
//   y[c,n] = x[c,n] - (sum_{w in taps, 0<=n-w<N} x[c,n-w]) / #{w in taps : 0<=n-w<N}
//
// Two kernels:
//   * filter_gather_kernel  - one thread per output, taps read from global/L2.  Any half-width,
//                             any size; used for tiny inputs and as the in-library cross-check.
//   * filter_stride_kernel  - the fast path.  A workgroup streams one channel-stretch through an
//                             LDS ring and keeps, per thread, the running tap sum S(n) of one
//                             residue class mod q:
//                                 S(n+q) = S(n) + sum_u d_q(u) * xz[n-u],  d_q(u) = tap(u+q) - tap(u)
//                             d_q is sparse because PARRM taps sit at near-multiples of the artefact
//                             period: with q ~ a multiple of the period the comb maps onto itself
//                             except where rounding moves a tooth edge and at the two ends.  That
//                             turns S=196 LDS reads + adds per output into n_delta (~20-30), which
//                             is what lets an f64 stencil approach the HBM roofline (SURVEY.md 7,
//                             hard part 1).  xz is x zero-padded outside [0, N), so the recurrence
//                             is exact at the recording edges too; only the divisor changes there.
#include <chrono>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "parrm_filter_internal.h"

using namespace parrm_filter;

namespace {

constexpr int kMaxRingLog2F64 = 14;  // 16384 * 8 B = 128 KiB of the 160 KiB LDS
constexpr int kMaxRowsPerFill = 8;
constexpr int kMaxBlock = 512;

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) filter_gather_kernel(FilterArgs a) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.out_len) return;
    const int64_t n = a.out_first + i;
    const TI *x = static_cast<const TI *>(a.x);
    (void)x;
    for (int64_t c = blockIdx.y; c < a.n_chans; c += gridDim.y) gather_output<TI, TO>(a, c, n);
}

// One workgroup = one (channel, stretch).  Threads r < q own residue r of the stretch; rows of q
// outputs are produced in lockstep, `rows_per_fill` rows per barrier.  D = padded number of delta
// taps; their byte offsets and weights live in VGPRs for the whole kernel (a scalar-memory load per
// tap inside the row loop serialises on lgkmcnt and was measured 15x slower).
template <typename TI, typename TO, int D>
__global__ void __launch_bounds__(kMaxBlock) filter_stride_kernel(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    TI *ring = reinterpret_cast<TI *>(lds_raw);

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int q = a.q;
    const int mask = a.ring_mask;
    const int G = a.rows_per_fill;
    const int hw = a.hw;
    constexpr int kEl = static_cast<int>(sizeof(TI));
    const int maskb = mask * kEl;  // byte mask for element-aligned addresses (mask = 2^k - 1)

    // the delta table is staged behind the ring so that each lane reads it into registers with
    // LDS loads (VGPR results) instead of scalar loads
    int32_t *tab_off = reinterpret_cast<int32_t *>(lds_raw + static_cast<size_t>(mask + 1) * kEl);
    double *tab_w = reinterpret_cast<double *>(tab_off + 128);
    for (int i = tid; i < D; i += nthr) {
        tab_off[i] = a.delta[i] * kEl;
        tab_w[i] = a.delta_w[i];
    }

    const int64_t blk = blockIdx.x;
    const int64_t c = blk / a.n_stretch;
    const int64_t st = blk - c * a.n_stretch;
    const int64_t s0 = a.out_first + st * a.stretch_len;
    int64_t s1 = s0 + a.stretch_len;
    if (s1 > a.out_first + a.out_len) s1 = a.out_first + a.out_len;
    const int len = static_cast<int>(s1 - s0);
    const TI *row = static_cast<const TI *>(a.x) + c * a.ldx;

    // prologue: ring <- xz[s0-hw, s0+(G+1)q+hw)
    int fill = (G + 1) * q + hw;  // exclusive front, relative to s0
    for (int rel = -hw + tid; rel < fill; rel += nthr) ring[rel & mask] = load_padded(a, row, s0 + rel);
    __syncthreads();

    int offb[D];
    double wv[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        offb[i] = tab_off[i];
        wv[i] = tab_w[i];
    }

    const bool active = tid < q;
    double S = 0.0;
    if (active) {  // full evaluation once per stretch: S(s0 + tid)
        double s_a = 0.0, s_b = 0.0;
        for (int r = 0; r < a.n_runs; ++r) {
            const int w_lo = a.runs[2 * r], w_hi = a.runs[2 * r + 1];
            int w = w_lo;
            for (; w + 1 <= w_hi; w += 2) {
                s_a += static_cast<double>(ring[(tid - w) & mask]);
                s_b += static_cast<double>(ring[(tid - w - 1) & mask]);
            }
            if (w <= w_hi) s_a += static_cast<double>(ring[(tid - w) & mask]);
        }
        S = s_a + s_b;
    }

    for (int m = 0; m * q < len; m += G) {
        // 1. issue the next fill's global loads; they land while the rows below are computed
        const bool more = (m + G) * q < len;  // wave-uniform
        TI pre[kMaxRowsPerFill];
        if (more && active) {
#pragma unroll
            for (int j = 0; j < kMaxRowsPerFill; ++j)
                if (j < G) pre[j] = load_padded(a, row, s0 + fill + j * q + tid);
        }
        // 2. G rows of outputs
        if (active) {
            for (int j = 0; j < G; ++j) {
                const int rel = (m + j) * q + tid;
                if ((m + j) * q >= len) break;
                if (rel < len) emit<TO, false>(a, c, s0 + rel, static_cast<double>(ring[rel & mask]), S);
                const int relb = rel * kEl;
                double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
                for (int i = 0; i < D; i += 4) {
                    const TI v0 = *reinterpret_cast<const TI *>(lds_raw + ((relb - offb[i]) & maskb));
                    const TI v1 = *reinterpret_cast<const TI *>(lds_raw + ((relb - offb[i + 1]) & maskb));
                    const TI v2 = *reinterpret_cast<const TI *>(lds_raw + ((relb - offb[i + 2]) & maskb));
                    const TI v3 = *reinterpret_cast<const TI *>(lds_raw + ((relb - offb[i + 3]) & maskb));
                    acc0 = fma(wv[i], static_cast<double>(v0), acc0);
                    acc1 = fma(wv[i + 1], static_cast<double>(v1), acc1);
                    acc2 = fma(wv[i + 2], static_cast<double>(v2), acc2);
                    acc3 = fma(wv[i + 3], static_cast<double>(v3), acc3);
                }
                S += (acc0 + acc1) + (acc2 + acc3);
            }
        }
        // 3. publish the prefetched samples (their slots alias data older than row m - see DESIGN.md)
        if (more) {
            if (active) {
#pragma unroll
                for (int j = 0; j < kMaxRowsPerFill; ++j)
                    if (j < G) ring[(fill + j * q + tid) & mask] = pre[j];
            }
            fill += G * q;
            __syncthreads();
        }
    }
}

}  // namespace

namespace {


constexpr int kDeltaSizes[] = {8, 16, 24, 32, 48, 64, 96, 128};

int padded_delta(int64_t n) {  // smallest compiled unroll >= n, 0 if none
    for (int d : kDeltaSizes)
        if (n <= d) return d;
    return 0;
}

int next_pow2_log2(int64_t v) {
    int l = 0;
    while ((int64_t{1} << l) < v) ++l;
    return l;
}

// Choose the recurrence stride: minimise (LDS reads per output row) / (lane utilisation).
void choose_stride(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *p,
                   std::vector<int32_t> *delta, std::vector<double> *weights) {
    double best_cost = 1e300;
    int64_t best_q = 0;
    auto tap_at = [&](int64_t w) -> int { return (w >= -hw && w <= hw) ? tap[w + hw] : 0; };
    std::vector<int64_t> taps_at;  // offsets of the taps, ascending
    for (int64_t w = -hw; w <= hw; ++w)
        if (tap[w + hw]) taps_at.push_back(w);
    for (int64_t q = 32; q <= kMaxBlock; ++q) {
        if (2 * hw + 3 * q > (int64_t{1} << kMaxRingLog2F64)) break;
        // |T xor (T - q)| = 2 (|T| - #{w in T : w + q in T})
        int64_t both = 0;
        for (const int64_t w : taps_at) both += tap_at(w + q);
        const int64_t nd = 2 * (static_cast<int64_t>(taps_at.size()) - both);
        const int pad = padded_delta(nd);
        if (pad == 0) continue;
        // Measured on MI355X (256 x 10 M f64, periods 1.3 ... 48 samples, strides 39 ... 507): the
        // kernel is latency-bound, time ~ (delta taps + ~26 instructions of per-row overhead) /
        // (waves resident per CU, saturating at about 8).  Residency: one power-of-two ring per
        // workgroup (LDS), ~40 + 3*pad VGPRs per lane (tap offsets and weights live in registers).
        const int64_t block = (q + 63) / 64 * 64;
        const int64_t ring_bytes = (int64_t{1} << next_pow2_log2(2 * hw + 3 * q)) * 8 + 2048;
        const int64_t by_lds = std::max<int64_t>(1, (160 * 1024) / ring_bytes);
        const int64_t by_regs = std::max<int64_t>(1, 512 / (40 + 3 * pad)) * 4;  // waves per CU
        const int64_t waves = std::min<int64_t>(std::min<int64_t>(by_lds, 8) * (block / 64), by_regs);
        const double cost = static_cast<double>(pad + 26) / static_cast<double>(std::min<int64_t>(waves, 8)) *
                            (0.75 + 0.25 * static_cast<double>(block) / static_cast<double>(q)) *
                            (1.0 + static_cast<double>(block) / 4096.0);  // past saturation smaller workgroups ran faster
        if (cost < best_cost - 1e-12) {  // ties: the smaller stride (measured: equal or better)
            best_cost = cost;
            best_q = q;
        }
    }
    if (const char *env = getenv("PARRM_STRIDE_Q")) {  // tuning knob: force the stride (must have a compiled unroll)
        const int64_t fq = atoll(env);
        if (fq >= 32 && fq <= kMaxBlock && 2 * hw + 3 * fq <= (int64_t{1} << kMaxRingLog2F64)) {
            int64_t both = 0;
            for (const int64_t w : taps_at) both += tap_at(w + fq);
            if (padded_delta(2 * (static_cast<int64_t>(taps_at.size()) - both)) != 0) best_q = fq;
        }
    }
    p->q = best_q;
    if (best_q == 0) return;  // half-width too large for the LDS ring (or no sparse d_q) -> gather kernel
    const int64_t q = best_q;
    for (int64_t u = -hw - q; u <= hw; ++u) {
        const int d = tap_at(u + q) - tap_at(u);
        if (d != 0) {
            delta->push_back(static_cast<int32_t>(u));
            weights->push_back(static_cast<double>(d));
        }
    }
    p->n_delta = static_cast<int64_t>(delta->size());
    p->n_delta_pad = padded_delta(p->n_delta);
    // padding entries re-read a real delta tap with weight 0: they add exactly 0 unless that sample
    // is non-finite, in which case the real tap poisons the sum anyway
    while (static_cast<int64_t>(delta->size()) < p->n_delta_pad) {
        delta->push_back((*delta)[0]);
        weights->push_back(0.0);
    }
    p->block_threads = static_cast<int>((q + 63) / 64 * 64);
    p->ring_log2_f64 = next_pow2_log2(2 * hw + 3 * q);
    int g = kMaxRowsPerFill;
    while (g > 1 && 2 * hw + (2 * g + 1) * q > (int64_t{1} << p->ring_log2_f64)) g >>= 1;
    p->rows_per_fill = g;
}

int resolve_kernel(const parrm_filter_plan *p, int64_t n_chans, int64_t out_len) {
    const bool has_phase = p->phase.n_groups > 0, has_stride = p->q != 0;
    switch (p->forced_kernel) {
        case PARRM_KERNEL_GATHER: return PARRM_KERNEL_GATHER;
        case PARRM_KERNEL_STRIDE: return has_stride ? PARRM_KERNEL_STRIDE : PARRM_KERNEL_GATHER;
        case PARRM_KERNEL_PHASE: return has_phase ? PARRM_KERNEL_PHASE : PARRM_KERNEL_GATHER;
        default: break;
    }
    // tiny problems: the per-stretch prologue (2*hw + ... samples) would dominate
    if (n_chans * out_len < (int64_t{1} << 15)) return PARRM_KERNEL_GATHER;
    if (has_phase) return PARRM_KERNEL_PHASE;
    if (has_stride) return PARRM_KERNEL_STRIDE;
    return p->segments.empty() ? PARRM_KERNEL_GATHER : PARRM_KERNEL_SEGMENTED;
}

void fill_plan_args(const parrm_filter_plan *plan, FilterArgs *a) {
    a->runs = plan->d_tables;
    a->tapcum = plan->d_tables + plan->off_tapcum;
    a->delta = plan->d_tables + plan->off_delta;
    a->n_runs = static_cast<int32_t>(plan->n_runs);
    a->delta_w = plan->d_weights;
    a->hw = static_cast<int32_t>(plan->hw);
    a->n_taps = static_cast<int32_t>(plan->n_taps);
    a->inv_taps = 1.0 / static_cast<double>(plan->n_taps);
    a->w_pos_min = static_cast<int32_t>(plan->w_pos_min);
    a->w_neg_min = static_cast<int32_t>(plan->w_neg_min);
}

// Last step of a segmented launch: acc holds the zero-padded tap sum of the WHOLE filter for every output;
// y = x - acc / (taps inside the recording), parrm.py:861-869.  A non-finite result is recomputed tap by tap
// (gather_output): the passes' running sums carry a NaN/Inf to the end of their stretches, far beyond the
// outputs whose taps reach the bad sample, and the direct evaluation zeroes exactly those.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) filter_combine_kernel(FilterArgs a, const double *acc, int64_t ldacc) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.out_len) return;
    const int64_t n = a.out_first + i;
    for (int64_t c = blockIdx.y; c < a.n_chans; c += gridDim.y) {
        const double xc = static_cast<double>(static_cast<const TI *>(a.x)[c * a.ldx + (n - a.buf_first)]);
        const double s = acc[c * ldacc + i];
        double y;
        if (n >= a.hw && n + a.hw < a.n_total) {
            y = xc - s * a.inv_taps;
        } else {
            const int v = valid_taps(a, n);
            y = v > 0 ? xc - s / static_cast<double>(v) : 0.0;
        }
        if (isfinite(y))
            static_cast<TO *>(a.y)[c * a.ldy + i] = static_cast<TO>(y);
        else
            gather_output<TI, TO>(a, c, n);
    }
}

template <typename TI, typename TO>
int launch_main(const parrm_filter_plan *p, FilterArgs *args, int kernel, hipStream_t stream);

// The generated kernel (parrm_filter_comb.hip) for this launch, or nullptr: float64 recordings, comb filters
// with a stride in [80, 176] whose taps sit within 40 residues of its multiples (as far as the LDS ring holds the halo).  PARRM_COMB=0 turns it off, PARRM_COMB=force takes it for any size; otherwise a
// launch must be large enough (2^25 samples) to be worth a possible hipRTC compile (seconds, once per filter
// geometry: code objects are cached on disk).  A failure to generate, compile or load is remembered and the
// launch falls back to filter_phase_kernel -- same results within the parity bar, never an error.
std::mutex g_comb_mutex;

// First use of a generated kernel: run it once on a small pseudo-random recording (a few stretches of a few ring
// periods, both ends of the recording included) and compare with the tap-by-tap gather kernel.  The generator is
// exercised by filters nobody has seen before (tap geometry is the caller's: create_filter's half-widths, omitted
// samples, direction); a kernel that does not reproduce the direct evaluation is never used.  ~1 ms, once per plan.
// `stream`: nullptr on the launching thread (first use of a cached kernel), a stream of its own on a worker thread.
template <typename TI, typename TO>
bool comb_self_test(const parrm_filter_plan *p, CombKernel *ck, hipStream_t stream = nullptr) {
    if (getenv("PARRM_COMB_NO_SELFTEST")) return true;
    const int64_t n = 2 * p->hw + 6 * comb_reach(ck) + 12345, c = 2;
    std::vector<TI> h(static_cast<size_t>(c * n));
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (TI &v : h) {  // xorshift, values in [-1, 1) with a full mantissa
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        v = static_cast<TI>(static_cast<double>(static_cast<int64_t>(st >> 11)) * (1.0 / 4503599627370496.0) - 1.0);
    }
    char *d = nullptr;
    bool ok = false;
    const size_t in_bytes = h.size() * sizeof(TI), out_bytes = h.size() * sizeof(TO);
    hipError_t e = hipMalloc(&d, in_bytes + 2 * out_bytes + 64);
    if (e == hipSuccess) {
        e = hipMemcpyAsync(d, h.data(), in_bytes, hipMemcpyHostToDevice, stream);
        FilterArgs a{};
        a.x = d;
        a.n_chans = c;
        a.plan_chans = c;
        a.buf_first = 0;
        a.buf_len = n;
        a.out_first = 0;
        a.out_len = n;
        a.n_total = n;
        a.ldx = n;
        a.ldy = n;
        fill_plan_args(p, &a);
        FilterArgs b = a;
        const size_t y_off = (in_bytes + 15) / 16 * 16;
        a.y = d + y_off;
        b.y = d + y_off + (out_bytes + 15) / 16 * 16;
        int rc = PARRM_OK;
        if (e == hipSuccess) rc = launch_comb(ck, &a, stream);
        if (e == hipSuccess && rc == PARRM_OK) rc = launch_main<TI, TO>(p, &b, PARRM_KERNEL_GATHER, stream);
        std::vector<TO> y1(h.size()), y2(h.size());
        if (e == hipSuccess && rc == PARRM_OK) e = hipMemcpyAsync(y1.data(), a.y, out_bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && rc == PARRM_OK) e = hipMemcpyAsync(y2.data(), b.y, out_bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && rc == PARRM_OK) e = hipStreamSynchronize(stream);
        if (e == hipSuccess && rc == PARRM_OK) {
            double worst = 0.0;
            for (size_t i = 0; i < y1.size(); ++i) {
                const double dd = std::fabs(static_cast<double>(y1[i]) - static_cast<double>(y2[i]));
                if (!(dd <= worst)) worst = dd;  // (NaN counts as a failure)
            }
            // samples are O(1): the direct evaluation rounds at ~1e-15; a float32 output is one rounding of each
            const double bar = sizeof(TO) == 8 ? 1e-11 : 3e-7;
            ok = worst <= bar;
            if (!ok) {
                char msg[128];
                snprintf(msg, sizeof msg, "self-test against the gather kernel failed (max |d| %.3e)", worst);
                comb_set_error(ck, msg);
            }
            // ... and a WINDOW call (the whole-recording case has buf_first = out_first = 0): the buffer
            // starts inside the recording and the outputs inside the buffer -- what the host-streamed and online
            // forms launch.  Every tap of every output lies in the buffer or beyond the recording's end.
            if (ok) {
                const int64_t bf = comb_reach(ck) + 13, of = bf + p->hw + 5;
                FilterArgs wa = a, wb = b;
                for (FilterArgs *w : {&wa, &wb}) {
                    w->x = d + static_cast<size_t>(bf) * sizeof(TI);
                    w->buf_first = bf;
                    w->buf_len = n - bf;
                    w->out_first = of;
                    w->out_len = n - of;
                }
                rc = launch_comb(ck, &wa, stream);
                if (rc == PARRM_OK) rc = launch_main<TI, TO>(p, &wb, PARRM_KERNEL_GATHER, stream);
                if (rc == PARRM_OK) e = hipMemcpyAsync(y1.data(), wa.y, out_bytes, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess && rc == PARRM_OK) e = hipMemcpyAsync(y2.data(), wb.y, out_bytes, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess && rc == PARRM_OK) e = hipStreamSynchronize(stream);
                if (e == hipSuccess && rc == PARRM_OK) {
                    double worst_w = 0.0;
                    for (int64_t ch = 0; ch < c; ++ch)
                        for (int64_t i = 0; i < n - of; ++i) {
                            const size_t at = static_cast<size_t>(ch * n + i);
                            const double dd = std::fabs(static_cast<double>(y1[at]) - static_cast<double>(y2[at]));
                            if (!(dd <= worst_w)) worst_w = dd;
                        }
                    ok = worst_w <= bar;
                    if (!ok) {
                        char msg[128];
                        snprintf(msg, sizeof msg, "self-test (window call) against the gather kernel failed (max |d| %.3e)", worst_w);
                        comb_set_error(ck, msg);
                    }
                } else {
                    ok = false;
                    comb_set_error(ck, "self-test (window call) could not run");
                }
            }
        } else {
            comb_set_error(ck, "self-test could not run");
        }
    } else {
        comb_set_error(ck, "self-test: out of device memory");
    }
    if (d) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(d);
    }
    return ok;
}

// Load (compile if need be), refuse spilling builds in favour of leaner ones, self-test: the kernel that may be used,
// or the last candidate with its error.  Runs on the launching thread (under g_comb_mutex) or on a plan's worker.
template <typename TI, typename TO>
int comb_build(const parrm_filter_plan *p, CombKernel *&ck, int64_t q, hipStream_t stream) {
    constexpr bool in32 = sizeof(TI) == 4, out32 = sizeof(TO) == 4;
    int state = -1;
    for (int attempt = 0; ck; ++attempt) {
        if (comb_load(ck)) {
            if (comb_self_test<TI, TO>(p, ck, stream)) state = 1;
            break;
        }
        CombKernel *next = strstr(comb_error(ck), "scratch") ? comb_generate(p->tap_mask, p->hw, q, attempt + 1, in32, out32) : nullptr;
        if (!next) break;
        comb_destroy(ck);
        ck = next;
    }
    if (ck && state != 1 && getenv("PARRM_COMB_VERBOSE"))
        fprintf(stderr, "parrm: generated filter kernel unavailable: %s\n", comb_error(ck));
    return state;
}

// a finished background build becomes the plan's kernel (g_comb_mutex held)
inline void comb_adopt(const parrm_filter_plan *p, int v) {
    std::shared_ptr<CombJob> &job = p->comb_job[v];
    if (p->comb_state[v] != 2 || !job || !job->done.load(std::memory_order_acquire)) return;
    if (job->worker.joinable()) job->worker.join();
    p->comb[v] = job->kernel;
    p->comb_state[v] = job->result_state;
    job.reset();
}

// variant: 0 = float64 -> float64, 1 = float32 -> float64, 2 = float32 -> float32
template <typename TI, typename TO>
const CombKernel *comb_for_launch(const parrm_filter_plan *p, const FilterArgs &a) {
    constexpr int v = sizeof(TI) == 8 ? 0 : (sizeof(TO) == 8 ? 1 : 2);
    constexpr bool in32 = sizeof(TI) == 4, out32 = sizeof(TO) == 4;
    static_assert(!(sizeof(TI) == 8 && sizeof(TO) == 4), "float64 recordings have float64 outputs");
    const char *env = getenv("PARRM_COMB");
    if (env && env[0] == '0') return nullptr;
    const bool force = env && env[0] == 'f';
    if (!force && a.plan_chans * a.out_len < (int64_t{1} << 25)) return nullptr;
    if (p->tap_mask.empty()) return nullptr;
    std::lock_guard<std::mutex> lock(g_comb_mutex);
    p->comb_last = v;
    CombKernel *&ck = p->comb[v];
    int &state = p->comb_state[v];
    if (state == 0) {
        const auto t_gen0 = std::chrono::steady_clock::now();
        const int64_t q = comb_search_stride(p->tap_mask, p->hw);
        ck = q ? comb_generate(p->tap_mask, p->hw, q, 0, in32, out32) : nullptr;
        if (getenv("PARRM_COMB_VERBOSE"))
            fprintf(stderr, "parrm: generated filter kernel: stride search + source in %.1f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gen0).count());
        // Unless forced, only where it has clearly less LDS traffic than the phase-major kernel (whose figure is
        // its number of delta taps): BASELINE geometry 17.8 reads per output against 28.
        if (ck && !force && p->phase.n_groups > 0 && comb_reads_per_output(ck) > 0.8 * (2.0 * p->phase.d_pad)) {
            comb_set_error(ck, "not used: no fewer LDS reads per output than the phase-major kernel");
            state = -1;
            return nullptr;
        }
        state = -1;
        // A geometry nobody has compiled yet costs ~1.7 s of hipRTC.  A plan in background mode
        // (parrm_filter_plan_set_background: the facade's unsharded filter_data; the reference's parameter explorer
        // re-filters on every widget event, _utils/_plotting.py:568-584) hands the build to a worker thread and
        // serves this and the following launches from the generic kernels -- same results within the parity bar --
        // until the kernel has passed its self-test.  PARRM_COMB=force and cached code objects stay synchronous.
        if (ck && p->comb_background && !force && !comb_code_cached(ck)) {
            auto job = std::make_shared<CombJob>();
            job->kernel = ck;
            ck = nullptr;
            state = 2;
            p->comb_job[v] = job;
            const int device = p->device;
            CombJob *raw = job.get();  // (the plan joins the worker before it goes away: parrm_filter_plan_destroy)
            job->worker = std::thread([p, raw, q, device]() {
                // One build at a time, at low priority: hipRTC is seconds of CPU on several threads, and three of them at
                // once starved the calling thread on a 16-core box (a filter_data call of 0.3 ms took 260-310 ms while
                // two earlier geometries were compiling -- measured, scripts/exp_background_compile.py).
                static std::mutex build_slot;
                (void)setpriority(PRIO_PROCESS, static_cast<id_t>(syscall(SYS_gettid)), 15);
                std::lock_guard<std::mutex> one_at_a_time(build_slot);
                hipStream_t st = nullptr;
                if (hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
                    raw->result_state = comb_build<TI, TO>(p, raw->kernel, q, st);
                    (void)hipStreamDestroy(st);
                } else if (raw->kernel) {
                    comb_set_error(raw->kernel, "background build: no device");
                }
                raw->done.store(1, std::memory_order_release);
            });
            return nullptr;
        }
        // a build that spills registers is refused at load time: leaner variants (smaller read batches) are tried
        state = comb_build<TI, TO>(p, ck, q, nullptr);
    }
    comb_adopt(p, v);
    if (state != 1 || !comb_accepts(ck, a)) return nullptr;
    return ck;
}

// A recurrence launch is followed by the repair pass (parrm_filter_internal.h: non-finite inputs).
template <typename TI, typename TO>
int launch_segmented(const parrm_filter_plan *p, FilterArgs a, hipStream_t stream) {
    // the float64 accumulator: the output itself when that is float64, a stream-ordered scratch otherwise
    double *acc = nullptr;
    int64_t ldacc = a.out_len;
    void *scratch = nullptr;
    if constexpr (sizeof(TO) == 8) {
        acc = static_cast<double *>(a.y);
        ldacc = a.ldy;
    } else {
        PARRM_HIP_CHECK(hipMallocAsync(&scratch, static_cast<size_t>(a.n_chans * a.out_len) * sizeof(double), stream));
        acc = static_cast<double *>(scratch);
    }
    int rc = PARRM_OK;
    hipError_t e = hipSuccess;  // (the first pass stores, the others add: no clear)
    for (size_t k = 0; k < p->segments.size() && e == hipSuccess && rc == PARRM_OK; ++k) {
        // window [c - H, c + H] of the taps = the sub-plan's filter applied to the recording shifted by c.  In the
        // frame of a recording of N + 2|c| samples that holds the data at [|c|, |c| + N) (the rest is not
        // addressable, i.e. reads as zero) every index is non-negative: output n sits at n - c + |c|.
        const parrm_filter_plan *sub = p->segments[k];
        const int64_t c = p->seg_centre[k], shift = c < 0 ? -c : c;
        FilterArgs b = a;
        fill_plan_args(sub, &b);
        b.y = acc;
        b.ldy = ldacc;
        b.buf_first = a.buf_first + shift;
        b.n_total = a.n_total + 2 * shift;
        b.out_first = a.out_first - c + shift;
        rc = launch_phase<TI, double>(sub, &b, stream, k == 0 ? 2 : 1);
    }
    if (e == hipSuccess && rc == PARRM_OK) {
        const int block = 256;
        const int64_t gx = (a.out_len + block - 1) / block;
        const int64_t gy = std::min<int64_t>(a.n_chans, 65535);
        hipLaunchKernelGGL((filter_combine_kernel<TI, TO>), dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)),
                           dim3(block), 0, stream, a, acc, ldacc);
        e = hipGetLastError();
    }
    if (scratch) {
        const hipError_t e2 = hipFreeAsync(scratch, stream);
        if (e == hipSuccess) e = e2;
    }
    if (rc != PARRM_OK) return rc;
    if (e != hipSuccess) return parrm::hip_fail(e, "filter: segmented launch");
    return PARRM_OK;
}

// parrm_filter_kernel_timing: HIP events right around the MAIN kernel of a launch (not the repair pass behind it),
// on the launch's own stream -- what bench.py's `roofline` is computed from.
struct KernelTiming {
    bool enabled = false, recorded = false;
    hipEvent_t start = nullptr, stop = nullptr;
};
thread_local KernelTiming *g_timing = nullptr;  // (a trivially destructible pointer: nothing runs at thread exit)

template <typename TI, typename TO>
int launch(const parrm_filter_plan *p, FilterArgs a, int kernel, hipStream_t stream) {
    if (kernel == PARRM_KERNEL_SEGMENTED) return launch_segmented<TI, TO>(p, a, stream);
    KernelTiming *t = g_timing && g_timing->enabled ? g_timing : nullptr;
    if (t) (void)hipEventRecord(t->start, stream);
    const int rc = launch_main<TI, TO>(p, &a, kernel, stream);
    if (t) t->recorded = hipEventRecord(t->stop, stream) == hipSuccess;
    if (rc != PARRM_OK || kernel == PARRM_KERNEL_GATHER || getenv("PARRM_NO_REPAIR_PASS")) return rc;
    const int64_t blocks = (a.n_chans * a.n_stretch + kRepairStretchesPerBlock - 1) / kRepairStretchesPerBlock;
    hipLaunchKernelGGL((filter_repair_kernel<TI, TO>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, a);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

template <typename TI, typename TO>
int launch_main(const parrm_filter_plan *p, FilterArgs *args, int kernel, hipStream_t stream) {
    FilterArgs &a = *args;
    if (kernel == PARRM_KERNEL_PHASE) {
        // The generated kernel: float64 recordings, and float32 recordings (widened to float64 on their way into the
        // ring, float64 arithmetic, a float32 output rounded once at the store).  For float32 -> float64 it is also
        // the faster kernel (6.9 ms against 9.0 on 256 ch x 10 M); for float32 -> float32 it is 6 % slower than the
        // packed phase-major kernel (6.35 against 5.99 ms) and 50 x closer to the exact evaluation (1.2e-7 of the
        // largest output against the packed sums' <= 6e-6) -- parity first: the default; PARRM_F32_PACKED=1 keeps the
        // packed kernel for large launches too.
        if constexpr (sizeof(TI) == 4 && sizeof(TO) == 4) {
            if (!getenv("PARRM_F32_PACKED"))
                if (const CombKernel *ck = comb_for_launch<TI, TO>(p, a)) return launch_comb(ck, args, stream);
        } else {
            if (const CombKernel *ck = comb_for_launch<TI, TO>(p, a)) return launch_comb(ck, args, stream);
        }
        return launch_phase<TI, TO>(p, args, stream);
    }
    if (kernel == PARRM_KERNEL_GATHER) {
        const int block = 256;
        const int64_t gx = (a.out_len + block - 1) / block;
        const int64_t gy = std::min<int64_t>(a.n_chans, 65535);
        PARRM_REQUIRE(gx <= 0x7fffffffLL, "filter: too many samples for one launch");
        hipLaunchKernelGGL((filter_gather_kernel<TI, TO>), dim3(static_cast<unsigned>(gx), static_cast<unsigned>(gy)),
                           dim3(block), 0, stream, a);
        PARRM_HIP_CHECK(hipGetLastError());
        return PARRM_OK;
    }
    // A filter the phase-major kernel does not take (wide teeth: its guard columns end at 6 residues) may still be one
    // the generated kernel takes (halo up to 40 columns, teeth as sliding sums) -- asked here when the stride kernel
    // was the plan's own choice, not a caller's (parrm_filter_plan_set_kernel).
    if (p->forced_kernel == PARRM_KERNEL_AUTO) {
        if constexpr (sizeof(TI) == 4 && sizeof(TO) == 4) {
            if (!getenv("PARRM_F32_PACKED"))
                if (const CombKernel *ck = comb_for_launch<TI, TO>(p, a)) return launch_comb(ck, args, stream);
        } else {
            if (const CombKernel *ck = comb_for_launch<TI, TO>(p, a)) return launch_comb(ck, args, stream);
        }
    }
    // stride kernel geometry
    const int q = static_cast<int>(p->q);
    const int G = p->rows_per_fill;
    int ring_log2 = p->ring_log2_f64;
    a.q = q;
    a.rows_per_fill = G;
    a.ring_mask = (1 << ring_log2) - 1;
    // stretch: ~64K samples, a whole number of fills; shrink while the grid would not fill the chip
    int64_t rows = std::max<int64_t>(G, (65536 / q) / G * G);
    auto blocks_for = [&](int64_t r) { return a.plan_chans * ((a.out_len + r * q - 1) / (r * q)); };
    while (rows > 4 * G && blocks_for(rows) < 2048) rows = std::max<int64_t>(G, (rows / 2) / G * G);
    a.stretch_len = rows * q;
    a.n_stretch = (a.out_len + a.stretch_len - 1) / a.stretch_len;
    const int64_t blocks = a.n_chans * a.n_stretch;
    PARRM_REQUIRE(blocks <= 0x7fffffffLL, "filter: too many workgroups for one launch");
    const size_t lds = (size_t{1} << ring_log2) * sizeof(TI) + 128 * sizeof(int32_t) + 128 * sizeof(double);
    a.n_delta_pad = static_cast<int32_t>(p->n_delta_pad);
    void (*kern)(FilterArgs) = nullptr;
    switch (p->n_delta_pad) {
        case 8: kern = filter_stride_kernel<TI, TO, 8>; break;
        case 16: kern = filter_stride_kernel<TI, TO, 16>; break;
        case 24: kern = filter_stride_kernel<TI, TO, 24>; break;
        case 32: kern = filter_stride_kernel<TI, TO, 32>; break;
        case 48: kern = filter_stride_kernel<TI, TO, 48>; break;
        case 64: kern = filter_stride_kernel<TI, TO, 64>; break;
        case 96: kern = filter_stride_kernel<TI, TO, 96>; break;
        case 128: kern = filter_stride_kernel<TI, TO, 128>; break;
        default: parrm::set_error("filter: no stride kernel for %lld delta taps", (long long)p->n_delta_pad); return PARRM_ERR_INVALID;
    }
    PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(p->block_threads), lds, stream, a);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

}  // namespace

extern "C" {

// Plan for a tap mask (tap[w + hw] = 1 for a tap at offset w; the centre may be one: segments of a long
// filter are masks of their own).  `allow_segments`: cut a filter that has no fast kernel into offset windows.
static int build_plan(const std::vector<int8_t> &tap, int64_t hw, bool allow_segments, parrm_filter_plan **plan);

int parrm_filter_plan_create(const double *h_filter, int64_t filter_len, parrm_filter_plan **plan) {
    PARRM_REQUIRE(h_filter && plan, "filter_plan_create: NULL argument");
    PARRM_REQUIRE(filter_len >= 3 && (filter_len & 1), "filter_plan_create: filter length must be odd and >= 3");
    PARRM_REQUIRE(filter_len < (int64_t{1} << 30), "filter_plan_create: filter too long");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        parrm::set_error("filter_plan_create: no HIP device is visible (there is no CPU fallback)");
        return PARRM_ERR_NO_DEVICE;
    }
    const int64_t hw = (filter_len - 1) / 2;
    std::vector<int8_t> tap(filter_len, 0);
    for (int64_t i = 0; i < filter_len; ++i) {
        if (i == hw) continue;  // the centre carries the 1 (parrm.py:831)
        if (h_filter[i] != 0.0) tap[i] = 1;
    }
    return build_plan(tap, hw, true, plan);
}

// Segmented plans.  A half-width the LDS cannot hold (2 hw + 3 q samples for the stride kernel, (2 hw / q + NG R)
// rows of q samples for the phase kernel) used to fall to the gather kernel: every output re-reads all its S taps
// from L2 -- 22 ms for 64 ch x 1 M at hw = 9 000 (722 taps), 134 ms at hw = 60 000.  The tap sum is additive over
// any partition of the taps, and a window [c - H, c + H] of offsets is a filter of half-width H applied to the
// recording shifted by c.  So the taps are cut into K windows that each HAVE a phase plan, each window's raw
// zero-padded tap sum is added into a float64 accumulator by one phase-kernel pass (MODE 1), and one light kernel
// turns the accumulator into outputs (own sample, divisor from the whole filter's cumulative tap counts).
// Traffic: 16 B (first pass: stores) + (K - 1) x 24 B + 24 B per sample instead of S x 8 B from L2.
static void build_segments(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *p) {
    double best_cost = 1e300;
    std::vector<parrm_filter_plan *> best;
    std::vector<int64_t> best_centre;
    auto release = [](std::vector<parrm_filter_plan *> &v) {
        for (parrm_filter_plan *q : v) (void)parrm_filter_plan_destroy(q);
        v.clear();
    };
    // Measured (64 ch x 1 M float64, scripts/sweep_segments.py): a pass costs ~0.36 ms + 0.055 ms per 1000 samples
    // of window half-width (the accumulator's 16 B per sample and the per-stretch prologues dominate), so few wide
    // windows win until the ring stops fitting (H ~ 8000 at q = 169).  Candidates: K equal windows for the few
    // smallest K whose windows can have phase plans.
    const char *force = getenv("PARRM_SEGMENT_HALFWIDTH");  // tuning knob: this window half-width only
    std::vector<int64_t> candidates;
    if (force) {
        candidates.push_back(atoll(force));
    } else {
        const int64_t span_all = 2 * hw + 1;
        for (int64_t k = 2; k <= 4096; k = std::max<int64_t>(k + 1, k * 9 / 8)) {
            const int64_t H = ((span_all + k - 1) / k) / 2 + 1;
            if (H > 9000) continue;
            if (H < 300) break;
            if (candidates.empty() || candidates.back() != H) candidates.push_back(H);
        }
    }
    int tried_ok = 0;
    for (const int64_t H : candidates) {
        if (H >= hw || H < 1) continue;
        if (tried_ok >= 3) break;  // costs only grow with more, narrower windows
        const int64_t width = 2 * H + 1, span = 2 * hw + 1;
        const int64_t k_windows = (span + width - 1) / width;
        std::vector<parrm_filter_plan *> subs;
        std::vector<int64_t> centres;
        double cost = 0.0;
        bool ok = true;
        for (int64_t k = 0; k < k_windows && ok; ++k) {
            const int64_t lo = -hw + k * width;                      // offsets [lo, lo + width)
            const int64_t c = lo + H;
            std::vector<int8_t> mask(width, 0);
            int64_t n = 0;
            for (int64_t j = 0; j < width; ++j) {
                const int64_t w = lo + j;
                if (w <= hw && tap[w + hw]) {
                    mask[j] = 1;
                    ++n;
                }
            }
            if (n == 0) continue;
            parrm_filter_plan *sub = nullptr;
            if (build_plan(mask, H, false, &sub) != PARRM_OK || sub->phase.n_groups == 0) {
                if (sub) (void)parrm_filter_plan_destroy(sub);
                ok = false;
                break;
            }
            subs.push_back(sub);
            centres.push_back(c);
            cost += 0.36 + 0.055 * static_cast<double>(H) / 1000.0;
        }
        if (ok && !subs.empty()) ++tried_ok;
        if (ok && !subs.empty() && cost < best_cost) {
            release(best);
            best_cost = cost;
            best.swap(subs);
            best_centre.swap(centres);
        }
        release(subs);
    }
    p->segments.swap(best);
    p->seg_centre.swap(best_centre);
}

static int build_plan(const std::vector<int8_t> &tap, int64_t hw, bool allow_segments, parrm_filter_plan **plan) {
    const int64_t filter_len = 2 * hw + 1;
    int64_t n_taps = 0;
    for (int64_t i = 0; i < filter_len; ++i) n_taps += tap[i];
    if (n_taps == 0) {
        parrm::set_error("filter_plan_create: the filter has no taps");
        return PARRM_ERR_EMPTY_FILTER;
    }
    auto *p = new parrm_filter_plan();
    p->hw = hw;
    p->n_taps = n_taps;
    for (int64_t w = 1; w <= hw; ++w) {  // tap[w + hw] is the tap at offset w: y[n] reads x[n - w]
        if (p->w_pos_min == 0 && tap[hw + w]) p->w_pos_min = w;
        if (p->w_neg_min == 0 && tap[hw - w]) p->w_neg_min = w;
    }
    std::vector<int32_t> runs;
    for (int64_t i = 0; i < filter_len;) {
        if (!tap[i]) {
            ++i;
            continue;
        }
        int64_t j = i;
        while (j + 1 < filter_len && tap[j + 1]) ++j;
        runs.push_back(static_cast<int32_t>(i - hw));
        runs.push_back(static_cast<int32_t>(j - hw));
        i = j + 1;
    }
    p->n_runs = static_cast<int64_t>(runs.size() / 2);
    std::vector<int32_t> tapcum(2 * hw + 2, 0);
    for (int64_t j = 0; j < filter_len; ++j) tapcum[j + 1] = tapcum[j] + tap[j];
    std::vector<int32_t> delta;
    std::vector<double> weights;
    choose_stride(tap, hw, p, &delta, &weights);
    weights.push_back(0.0);
    std::vector<int32_t> phase_tab;
    plan_phase(tap, hw, p, &phase_tab);
    phase_tab.push_back(0);

    std::vector<int32_t> tables(runs);
    p->off_tapcum = static_cast<int64_t>(tables.size());
    tables.insert(tables.end(), tapcum.begin(), tapcum.end());
    p->off_delta = static_cast<int64_t>(tables.size());
    tables.insert(tables.end(), delta.begin(), delta.end());
    tables.push_back(0);
    // one device allocation: weights | tables | phase table
    const size_t wb = weights.size() * sizeof(double), tb = (tables.size() * sizeof(int32_t) + 7) & ~size_t{7},
                 pb = phase_tab.size() * sizeof(int32_t);
    std::vector<unsigned char> blob(wb + tb + pb);
    std::memcpy(blob.data(), weights.data(), wb);
    std::memcpy(blob.data() + wb, tables.data(), tables.size() * sizeof(int32_t));
    std::memcpy(blob.data() + wb + tb, phase_tab.data(), pb);
    hipError_t e = hipGetDevice(&p->device);
    void *d_blob = nullptr;
    if (e == hipSuccess) e = hipMalloc(&d_blob, blob.size());
    if (e == hipSuccess) e = hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d_blob) (void)hipFree(d_blob);
        delete p;
        return parrm::hip_fail(e, "filter_plan_create: table upload");
    }
    p->d_weights = static_cast<double *>(d_blob);
    p->d_tables = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(d_blob) + wb);
    p->d_phase_tab = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(d_blob) + wb + tb);
    if (allow_segments && p->phase.n_groups == 0 && p->q == 0) build_segments(tap, hw, p);
    if (allow_segments) p->tap_mask = tap;  // (sub-plans of a segmented filter never take the generated kernel)
    *plan = p;
    return PARRM_OK;
}

int parrm_filter_plan_destroy(parrm_filter_plan *plan) {
    if (!plan) return PARRM_OK;
    if (plan->d_weights) (void)hipFree(plan->d_weights);  // the one allocation (weights | tables | phase table)
    for (parrm_filter_plan *sub : plan->segments) (void)parrm_filter_plan_destroy(sub);
    for (int v = 0; v < 3; ++v)
        if (plan->comb_job[v]) {  // a background build still uses the plan's tables: wait for it
            if (plan->comb_job[v]->worker.joinable()) plan->comb_job[v]->worker.join();
            parrm_filter::comb_destroy(plan->comb_job[v]->kernel);
            plan->comb_job[v].reset();
        }
    for (parrm_filter::CombKernel *ck : plan->comb) parrm_filter::comb_destroy(ck);
    delete plan;
    return PARRM_OK;
}

int parrm_filter_plan_query(const parrm_filter_plan *plan, parrm_filter_plan_info *info) {
    PARRM_REQUIRE(plan && info, "filter_plan_query: NULL argument");
    info->half_width = plan->hw;
    info->n_taps = plan->n_taps;
    info->n_runs = plan->n_runs;
    info->stride = plan->q;
    info->n_delta = plan->n_delta;
    info->ring_len = plan->q ? (int64_t{1} << plan->ring_log2_f64) : 0;
    info->rows_per_fill = plan->rows_per_fill;
    info->block_threads = plan->block_threads;
    info->kernel = resolve_kernel(plan, int64_t{1} << 20, int64_t{1} << 20);
    info->phase_stride = plan->phase.n_groups ? plan->phase.q : 0;
    info->phase_delta = plan->phase.n_groups ? 2 * plan->phase.d_pad : 0;
    info->phase_guard = plan->phase.guard;
    info->phase_groups = plan->phase.n_groups;
    info->phase_rows = plan->phase.rows;
    info->phase_row_slots = plan->phase.m_slots;
    info->phase_residues = 1;
    info->reserved = static_cast<int32_t>(plan->segments.size());  // passes of a segmented plan (0: not segmented)
    return PARRM_OK;
}

int parrm_filter_plan_generated(const parrm_filter_plan *plan, int *state, int *stride, char *message, size_t message_len) {
    PARRM_REQUIRE(plan && state, "filter_plan_generated: NULL argument");
    std::lock_guard<std::mutex> lock(g_comb_mutex);
    const int v = plan->comb_last;  // (the element types of the last launch that asked for a generated kernel)
    comb_adopt(plan, v);
    *state = plan->comb_state[v];
    if (stride) *stride = plan->comb[v] ? comb_stride(plan->comb[v]) : 0;
    if (message && message_len)
        snprintf(message, message_len, "%s", *state == 2 ? "being compiled in the background" : plan->comb[v] ? comb_error(plan->comb[v]) : "");
    return PARRM_OK;
}

int parrm_filter_plan_set_background(parrm_filter_plan *plan, int on) {
    PARRM_REQUIRE(plan, "filter_plan_set_background: NULL plan");
    std::lock_guard<std::mutex> lock(g_comb_mutex);
    plan->comb_background = on != 0;
    return PARRM_OK;
}

int parrm_filter_kernel_timing(int enable, float *last_ms) {
    if (!g_timing) g_timing = new KernelTiming();
    KernelTiming &t = *g_timing;
    if (last_ms) {
        *last_ms = -1.0f;
        if (t.recorded) {
            PARRM_HIP_CHECK(hipEventSynchronize(t.stop));
            PARRM_HIP_CHECK(hipEventElapsedTime(last_ms, t.start, t.stop));
        }
    }
    if (enable && !t.start) {
        PARRM_HIP_CHECK(hipEventCreate(&t.start));
        PARRM_HIP_CHECK(hipEventCreate(&t.stop));
    }
    if (!enable && t.start) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
        t.start = t.stop = nullptr;
        t.recorded = false;
    }
    t.enabled = enable != 0;
    return PARRM_OK;
}

int parrm_filter_plan_set_kernel(parrm_filter_plan *plan, int kernel) {
    PARRM_REQUIRE(plan, "filter_plan_set_kernel: NULL plan");
    PARRM_REQUIRE(kernel >= PARRM_KERNEL_AUTO && kernel <= PARRM_KERNEL_PHASE,
                  "filter_plan_set_kernel: unknown kernel %d", kernel);
    PARRM_REQUIRE(kernel != PARRM_KERNEL_PHASE || plan->phase.n_groups != 0,
                  "filter_plan_set_kernel: this filter has no phase-major plan (period too short or taps too spread)");
    PARRM_REQUIRE(kernel != PARRM_KERNEL_STRIDE || plan->q != 0,
                  "filter_plan_set_kernel: half-width %lld does not fit the LDS ring", (long long)plan->hw);
    plan->forced_kernel = kernel;
    return PARRM_OK;
}

int parrm_filter_apply_window(const parrm_filter_plan *plan, const void *d_x, int x_dtype, void *d_y,
                              int y_dtype, int64_t n_chans, int64_t buf_first, int64_t buf_len,
                              int64_t out_first, int64_t out_len, int64_t n_total, int64_t ldx,
                              int64_t ldy, void *stream) {
    return parrm_filter_apply_block(plan, d_x, x_dtype, d_y, y_dtype, n_chans, n_chans, buf_first, buf_len, out_first,
                                    out_len, n_total, ldx, ldy, stream);
}

int parrm_filter_apply_block(const parrm_filter_plan *plan, const void *d_x, int x_dtype, void *d_y, int y_dtype,
                             int64_t n_chans, int64_t total_chans, int64_t buf_first, int64_t buf_len,
                             int64_t out_first, int64_t out_len, int64_t n_total, int64_t ldx, int64_t ldy,
                             void *stream) {
    PARRM_REQUIRE(total_chans >= n_chans, "filter_apply: a channel block cannot be larger than its recording");
    PARRM_REQUIRE(plan, "filter_apply: NULL plan");
    PARRM_REQUIRE(n_chans >= 0 && out_len >= 0 && n_total >= 0 && buf_len >= 0, "filter_apply: negative size");
    if (n_chans == 0 || out_len == 0) return PARRM_OK;
    PARRM_REQUIRE(d_x && d_y, "filter_apply: NULL data pointer");
    PARRM_REQUIRE(x_dtype == PARRM_F32 || x_dtype == PARRM_F64, "filter_apply: bad x_dtype %d", x_dtype);
    PARRM_REQUIRE(y_dtype == PARRM_F64 || (y_dtype == PARRM_F32 && x_dtype == PARRM_F32),
                  "filter_apply: y_dtype must be f64, or f32 for f32 input");
    PARRM_REQUIRE(out_first >= 0 && out_first + out_len <= n_total, "filter_apply: outputs outside the recording");
    PARRM_REQUIRE(ldx >= buf_len && ldy >= out_len, "filter_apply: row stride smaller than the row");
    const int64_t need_lo = std::max<int64_t>(out_first - plan->hw, 0);
    const int64_t need_hi = std::min<int64_t>(out_first + out_len + plan->hw, n_total);
    PARRM_REQUIRE(buf_first <= need_lo && buf_first + buf_len >= need_hi,
                  "filter_apply: window [%lld,%lld) does not cover the halo [%lld,%lld)", (long long)buf_first,
                  (long long)(buf_first + buf_len), (long long)need_lo, (long long)need_hi);
    // the plan's tables are a hipMalloc on the device it was created on: a launch from another device
    // would dereference that device's memory without peer access
    int current = -1;
    PARRM_HIP_CHECK(hipGetDevice(&current));
    PARRM_REQUIRE(current == plan->device, "filter_apply: the plan was built on device %d but device %d is current",
                  plan->device, current);

    FilterArgs a{};
    a.x = d_x;
    a.y = d_y;
    a.n_chans = n_chans;
    a.plan_chans = total_chans;
    a.buf_first = buf_first;
    a.buf_len = buf_len;
    a.out_first = out_first;
    a.out_len = out_len;
    a.n_total = n_total;
    a.ldx = ldx;
    a.ldy = ldy;
    fill_plan_args(plan, &a);
    const int kernel = resolve_kernel(plan, total_chans, out_len);
    hipStream_t s = parrm::as_stream(stream);
    if (x_dtype == PARRM_F64) return launch<double, double>(plan, a, kernel, s);
    if (y_dtype == PARRM_F64) return launch<float, double>(plan, a, kernel, s);
    return launch<float, float>(plan, a, kernel, s);
}

int parrm_filter_apply(const parrm_filter_plan *plan, const void *d_x, int x_dtype, void *d_y, int y_dtype,
                       int64_t n_chans, int64_t n_samples, int64_t ldx, int64_t ldy, void *stream) {
    return parrm_filter_apply_window(plan, d_x, x_dtype, d_y, y_dtype, n_chans, 0, n_samples, 0, n_samples,
                                     n_samples, ldx, ldy, stream);
}

}  // extern "C"
