// filter_data fast path for comb-like filters (artefact period >= ~64 samples): the "phase-major"
// kernel.  Same recurrence as filter_stride_kernel,
//
//     S(n+q) = S(n) + sum_u d_q(u) xz[n-u],        d_q(u) = tap(u+q) - tap(u),
//
// but laid out so that the work per tap is amortised:
//
//   * A sample at offset rel = m*q + r from the stretch start (row m, residue r) lives in LDS at
//     buf[(r + B) * RS + slot(m)]: residue-major, rows along the fast axis.  A delta tap u = a*q + b
//     (|b| <= B because the comb teeth sit near multiples of q) is read by thread r at residue
//     r - b, row m - a.  For R consecutive rows of the SAME thread those are R adjacent elements:
//     one address (lane part + scalar part) and R ds_reads with immediate offsets.
//   * The scalar part (row slot, wrap, -b*RS) is wave-uniform and is computed on the scalar unit.
//   * B mirrored residues on each side of a row absorb r - b < 0 / >= q, and R-1 mirrored head
//     slots absorb slot + i >= M, so no lane ever needs a wrap.
//   * Row sums Delta_m do not depend on S, so NG groups of waves work on different rows of the
//     same ring (NG x the waves per LDS byte) and only exchange one partial sum per thread per
//     iteration through LDS.
//
// Rows outside the recording are zero-filled, so the recurrence is exact at the edges; only the
// divisor (number of valid taps) changes there (emit()).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "parrm_filter_internal.h"

namespace parrm_filter {

namespace {

constexpr int kMaxGuard = 24;
constexpr size_t kLdsTwoBlocks = 80 * 1024;   // 2 workgroups per CU
constexpr size_t kLdsOneBlock = 160 * 1024;

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~size_t{15}; }

// LDS byte offset of a __shared__ object (address space 3 pointers are 32-bit offsets)
__device__ inline unsigned lds_offset(const void *p) {
    return static_cast<unsigned>(reinterpret_cast<size_t>(
        (const __attribute__((address_space(3))) void *)p));
}

// ---- tap reads -------------------------------------------------------------------------------
// A batch is 2*TB row-sets (TB taps of each sign) of R consecutive elements: 2*TB*R separate
// full-rate ds_read_b64 / ds_read_b32 with immediate offsets.  They go through inline asm because
// hipcc would pair adjacent-row reads into ds_read2_b64, which moves 16 B/lane at HALF the
// ds_read_b64 rate (MI355X_MICROARCH.md, LDS table).  The reads are software-pipelined: the asm
// that issues batch k also waits, with a COUNTED lgkmcnt, for batch k-1 -- LDS operations return
// in order, so "at most n outstanding" right after issuing n reads means everything older is
// back, whatever else (LDS or scalar loads) the compiler had in flight before the statement.
// The LDS pipe therefore never drains inside the tap loop (draining every batch measured ~45 %
// of the LDS rate at 3 waves/SIMD).  Uses of a batch are kept below its wait by a
// sched_barrier (the compiler does not know the asm results are asynchronous).
template <int R>
struct TapBatch {
    static constexpr int TB = R <= 3 ? 2 : 1;  // taps per sign per batch; 2*TB*R <= 12 (lgkmcnt is 4 bits)
    static constexpr int N = 2 * TB * R;
};

#define PARRM_RD3(op, o0, o1, o2, a, s) \
    op " %" #o0 ", %" #a "\n\t" op " %" #o1 ", %" #a " offset:" #s "\n\t" op " %" #o2 ", %" #a " offset:2*" #s "\n\t"
#define PARRM_RD2(op, o0, o1, a, s) op " %" #o0 ", %" #a "\n\t" op " %" #o1 ", %" #a " offset:" #s "\n\t"
#define PARRM_RD4(op, o0, o1, o2, o3, a, s)                                                              \
    op " %" #o0 ", %" #a "\n\t" op " %" #o1 ", %" #a " offset:" #s "\n\t" op " %" #o2 ", %" #a " offset:2*" #s \
       "\n\t" op " %" #o3 ", %" #a " offset:3*" #s "\n\t"

// v[j][i]: row-set j (0..2*TB-1), row i.  WAIT: also wait for everything issued before this batch.
template <typename TI, int R, bool WAIT>
__device__ inline void tap_batch(const unsigned (&ad)[2 * TapBatch<R>::TB], TI (&v)[2 * TapBatch<R>::TB][R]) {
    if constexpr (sizeof(TI) == 8 && R == 3) {
#define PARRM_BODY(w)                                                                                       \
    asm volatile(PARRM_RD3("ds_read_b64", 0, 1, 2, 12, 8) PARRM_RD3("ds_read_b64", 3, 4, 5, 13, 8)           \
                     PARRM_RD3("ds_read_b64", 6, 7, 8, 14, 8) PARRM_RD3("ds_read_b64", 9, 10, 11, 15, 8) w   \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[1][0]), "=&v"(v[1][1]),           \
                   "=&v"(v[1][2]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[2][2]), "=&v"(v[3][0]),           \
                   "=&v"(v[3][1]), "=&v"(v[3][2])                                                           \
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(12)"); else PARRM_BODY("");
#undef PARRM_BODY
    } else if constexpr (sizeof(TI) == 4 && R == 3) {
#define PARRM_BODY(w)                                                                                       \
    asm volatile(PARRM_RD3("ds_read_b32", 0, 1, 2, 12, 4) PARRM_RD3("ds_read_b32", 3, 4, 5, 13, 4)           \
                     PARRM_RD3("ds_read_b32", 6, 7, 8, 14, 4) PARRM_RD3("ds_read_b32", 9, 10, 11, 15, 4) w   \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[1][0]), "=&v"(v[1][1]),           \
                   "=&v"(v[1][2]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[2][2]), "=&v"(v[3][0]),           \
                   "=&v"(v[3][1]), "=&v"(v[3][2])                                                           \
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(12)"); else PARRM_BODY("");
#undef PARRM_BODY
    } else if constexpr (sizeof(TI) == 8 && R == 2) {
#define PARRM_BODY(w)                                                                                     \
    asm volatile(PARRM_RD2("ds_read_b64", 0, 1, 8, 8) PARRM_RD2("ds_read_b64", 2, 3, 9, 8)                  \
                     PARRM_RD2("ds_read_b64", 4, 5, 10, 8) PARRM_RD2("ds_read_b64", 6, 7, 11, 8) w          \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]),          \
                   "=&v"(v[2][1]), "=&v"(v[3][0]), "=&v"(v[3][1])                                          \
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(8)"); else PARRM_BODY("");
#undef PARRM_BODY
    } else if constexpr (sizeof(TI) == 4 && R == 2) {
#define PARRM_BODY(w)                                                                                     \
    asm volatile(PARRM_RD2("ds_read_b32", 0, 1, 8, 4) PARRM_RD2("ds_read_b32", 2, 3, 9, 4)                  \
                     PARRM_RD2("ds_read_b32", 4, 5, 10, 4) PARRM_RD2("ds_read_b32", 6, 7, 11, 4) w          \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]),          \
                   "=&v"(v[2][1]), "=&v"(v[3][0]), "=&v"(v[3][1])                                          \
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(8)"); else PARRM_BODY("");
#undef PARRM_BODY
    } else if constexpr (sizeof(TI) == 8 && R == 4) {
#define PARRM_BODY(w)                                                                                        \
    asm volatile(PARRM_RD4("ds_read_b64", 0, 1, 2, 3, 8, 8) PARRM_RD4("ds_read_b64", 4, 5, 6, 7, 9, 8) w       \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]),            \
                   "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])                                            \
                 : "v"(ad[0]), "v"(ad[1]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(8)"); else PARRM_BODY("");
#undef PARRM_BODY
    } else {
        static_assert(sizeof(TI) == 4 && R == 4, "unsupported tap batch shape");
#define PARRM_BODY(w)                                                                                        \
    asm volatile(PARRM_RD4("ds_read_b32", 0, 1, 2, 3, 8, 4) PARRM_RD4("ds_read_b32", 4, 5, 6, 7, 9, 4) w       \
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]),            \
                   "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])                                            \
                 : "v"(ad[0]), "v"(ad[1]))
        if constexpr (WAIT) PARRM_BODY("s_waitcnt lgkmcnt(8)"); else PARRM_BODY("");
#undef PARRM_BODY
    }
}

template <typename TI, typename TO, int DP, int R>
__global__ void __launch_bounds__(1024) filter_phase_kernel(FilterArgs a, PhaseGeom p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int kEl = static_cast<int>(sizeof(TI));
    const int q = p.q, QP = p.qp, B = p.guard, M = p.m_slots, RS = p.rs, NG = p.n_groups;
    const int a_lo = p.a_lo, a_hi = p.a_hi;
    const int NGR = NG * R;
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid / QP);  // whole waves per group
    const int r = tid - g * QP;
    const bool active = r < q;
    const int n_res = q + 2 * B;
    double *xchg = reinterpret_cast<double *>(lds_raw + align16(static_cast<size_t>(n_res) * RS * kEl));

    // wave-uniform tap tables (scalar registers)
    int tp_a[DP], tp_c[DP], tm_a[DP], tm_c[DP];
#pragma unroll
    for (int t = 0; t < DP; ++t) {
        tp_a[t] = p.tab[t];
        tp_c[t] = -p.tab[DP + t] * RS * kEl;
        tm_a[t] = p.tab[2 * DP + t];
        tm_c[t] = -p.tab[3 * DP + t] * RS * kEl;
    }
    const int32_t *runs = p.tab + 4 * DP;

    const int64_t blk = blockIdx.x;
    const int64_t c = blk / a.n_stretch;
    const int64_t st = blk - c * a.n_stretch;
    const int64_t s0 = a.out_first + st * a.stretch_len;
    int64_t s1 = s0 + a.stretch_len;
    if (s1 > a.out_first + a.out_len) s1 = a.out_first + a.out_len;
    const int len = static_cast<int>(s1 - s0);
    const int rows_total = (len + q - 1) / q;
    const TI *xrow = static_cast<const TI *>(a.x) + c * a.ldx;

    // byte offset of (this residue, slot 0); idle lanes of the last wave shadow residue 0
    const int lane_base = ((active ? r : 0) + B) * RS * kEl;
    // slot(row) = (row + a_hi) mod M
    auto store_cells = [&](int row, int sl, int rr, TI v) {
        // main cell (+ mirrored head so that slot + i, i < R, never wraps)
        TI *cell = reinterpret_cast<TI *>(lds_raw + static_cast<size_t>((rr + B) * RS + sl) * kEl);
        *cell = v;
        if (sl < R - 1) cell[M] = v;
        if (rr >= q - B) {  // also residue rr - q of the next row
            int s2 = sl + 1;
            if (s2 >= M) s2 -= M;
            TI *gc = reinterpret_cast<TI *>(lds_raw + static_cast<size_t>((rr - q + B) * RS + s2) * kEl);
            *gc = v;
            if (s2 < R - 1) gc[M] = v;
        }
        if (rr < B && row > -a_hi) {  // also residue rr + q of the previous row
            int s2 = sl - 1;
            if (s2 < 0) s2 += M;
            TI *gc = reinterpret_cast<TI *>(lds_raw + static_cast<size_t>((rr + q + B) * RS + s2) * kEl);
            *gc = v;
            if (s2 < R - 1) gc[M] = v;
        }
    };

    // prologue: rows [-a_hi, NGR - a_lo] of xz.  Eight loads per thread are issued before the first
    // one is consumed; a load->store-per-element loop exposes one HBM latency per element.
    {
        const int n_rows0 = NGR - a_lo + a_hi + 1;
        const int nthr = blockDim.x;
        const int total = n_rows0 * QP;
        for (int base = 0; base < total; base += 8 * nthr) {
            TI v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + j * nthr + tid;
                const int k = idx / QP, rr = idx - k * QP;
                v[j] = (idx < total && rr < q) ? load_padded(a, xrow, s0 + static_cast<int64_t>(k - a_hi) * q + rr) : TI(0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + j * nthr + tid;
                const int k = idx / QP, rr = idx - k * QP;
                if (idx < total && rr < q) store_cells(k - a_hi, k % M, rr, v[j]);
            }
        }
    }
    __syncthreads();

    // S at row 0 for this residue: full evaluation, once per stretch (both groups, redundantly)
    double S = 0.0;
    if (active) {
        double s_a = 0.0, s_b = 0.0;
        for (int k = 0; k < p.n_runs; ++k) {
            const int ra = runs[3 * k], b_lo = runs[3 * k + 1], b_hi = runs[3 * k + 2];
            int sl = (0 - a_lo - ra + a_hi) % M;  // slot(0 - a), a = ra + a_lo
            const unsigned char *base = lds_raw + lane_base + sl * kEl;
            int b = b_lo;
            for (; b + 1 <= b_hi; b += 2) {
                s_a += static_cast<double>(*reinterpret_cast<const TI *>(base - b * RS * kEl));
                s_b += static_cast<double>(*reinterpret_cast<const TI *>(base - (b + 1) * RS * kEl));
            }
            if (b <= b_hi) s_a += static_cast<double>(*reinterpret_cast<const TI *>(base - b * RS * kEl));
        }
        S = s_a + s_b;
    }

    // scalar slot bookkeeping for this wave's first row m = mk + g*R
    int sl_top = (g * R - a_lo + a_hi) % M;        // slot(m - a_lo): taps count down from here
    int sl_own = (g * R + a_hi) % M;               // slot(m)
    int sl_fill = (NGR - a_lo + 1 + g * R + a_hi) % M;  // slot of this wave's first prefetched row
    int par = 0;

    // rows of iteration 1 are requested now (they are published during iteration 0); every
    // iteration then requests the rows of iteration k+2, so a load has a whole iteration to land
    TI pre_a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t n1 = s0 + static_cast<int64_t>(NGR - a_lo + 1 + g * R + i) * q + r;
        pre_a[i] = (active && NGR < rows_total) ? load_padded(a, xrow, n1) : TI(0);
    }

    // One iteration = NG*R rows.  `pub` holds the rows requested one iteration ago (published to LDS
    // here), `req` receives the rows of iteration k+2.  The loop below alternates two register sets so
    // that no copy ever touches a register with a load in flight (a copy would force vmcnt(0)).
    auto iteration = [&](const int mk, TI (&pub)[R], TI (&req)[R]) {
        const int m = mk + g * R;
        const bool more = mk + NGR < rows_total;
        const bool more2 = mk + 2 * NGR < rows_total;
        // 1. request this thread's share of the rows iteration k+2 needs
        const int frow = mk + NGR - a_lo + 1 + g * R;
#pragma unroll
        for (int i = 0; i < R; ++i)
            req[i] = (more2 && active && !(p.debug & 4))
                           ? load_padded(a, xrow, s0 + static_cast<int64_t>(frow + NGR + i) * q + r)
                           : TI(0);
        // 2. row sums Delta_i = sum_u d_q(u) xz[(m+i)q + r - u]
        double accp[R], accm[R];
        TI xo[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            accp[i] = 0.0;
            accm[i] = 0.0;
            xo[i] = *reinterpret_cast<const TI *>(lds_raw + lane_base + (sl_own + i) * kEl);
        }
        const unsigned lds0 = lds_offset(lds_raw) + static_cast<unsigned>(lane_base);
        if (!(p.debug & 1)) {
            constexpr int TB = TapBatch<R>::TB;
            constexpr int NB = DP / TB;
            static_assert(DP % TB == 0, "DP must be a multiple of the batch size");
            TI v[2][2 * TB][R];
            auto addresses = [&](int k, unsigned (&ad)[2 * TB]) {
#pragma unroll
                for (int t = 0; t < TB; ++t) {
                    int sp = sl_top - tp_a[k * TB + t];
                    sp += (sp >> 31) & M;
                    ad[2 * t] = lds0 + static_cast<unsigned>(sp * kEl + tp_c[k * TB + t]);
                    int sm = sl_top - tm_a[k * TB + t];
                    sm += (sm >> 31) & M;
                    ad[2 * t + 1] = lds0 + static_cast<unsigned>(sm * kEl + tm_c[k * TB + t]);
                }
            };
            auto accumulate = [&](TI (&b)[2 * TB][R]) {
#pragma unroll
                for (int t = 0; t < TB; ++t)
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        accp[i] += static_cast<double>(b[2 * t][i]);
                        accm[i] += static_cast<double>(b[2 * t + 1][i]);
                    }
            };
            unsigned ad[2 * TB];
            addresses(0, ad);
            tap_batch<TI, R, false>(ad, v[0]);
#pragma unroll
            for (int k = 1; k < NB; ++k) {
                addresses(k, ad);
                tap_batch<TI, R, true>(ad, v[k & 1]);  // issues batch k, returns once batch k-1 is back
                __builtin_amdgcn_sched_barrier(0);
                accumulate(v[(k - 1) & 1]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            accumulate(v[(NB - 1) & 1]);
        }
        double delta[R];
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            delta[i] = accp[i] - accm[i];
            tot += delta[i];
        }
        xchg[(par * NG + g) * QP + r] = tot;
        // 3. publish the rows requested one iteration ago (their slots alias rows older than mk - a_hi)
        if (more && active && !(p.debug & 4)) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int sl = sl_fill + i;
                if (sl >= M) sl -= M;
                store_cells(frow + i, sl, r, pub[i]);
            }
        }
        // LDS-only barrier: a plain __syncthreads() also waits vmcnt(0), i.e. for the global loads
        // requested above and for the previous iteration's output stores, once per iteration
        if (!(p.debug & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // 4. chain the groups: S at this wave's first row, and S at the next iteration's base row
        double s_row = S, s_all = 0.0;
        for (int gg = 0; gg < NG; ++gg) {
            const double tg = xchg[(par * NG + gg) * QP + r];
            if (gg < g) s_row += tg;
            s_all += tg;
        }
        S += s_all;
        // 5. outputs
        if (active && !(p.debug & 2)) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int rel = (m + i) * q + r;
                if (rel < len) emit<TO>(a, c, s0 + rel, static_cast<double>(xo[i]), s_row);
                s_row += delta[i];
            }
        }
        par ^= 1;
        sl_top += NGR;
        if (sl_top >= M) sl_top -= M;
        sl_own += NGR;
        if (sl_own >= M) sl_own -= M;
        sl_fill += NGR;
        if (sl_fill >= M) sl_fill -= M;
    };
    TI pre_b[R];
    for (int mk = 0; mk < rows_total; mk += 2 * NGR) {
        iteration(mk, pre_a, pre_b);
        if (mk + NGR < rows_total) iteration(mk + NGR, pre_b, pre_a);
    }
}

int pad_half(int64_t n) {
    for (int d : {4, 8, 12, 14, 16, 20, 24, 32})
        if (n <= d) return d;
    return 0;
}

struct Split {
    int a, b;
};
inline Split split(int64_t u, int64_t q) {
    const int64_t a = static_cast<int64_t>(std::floor(static_cast<double>(u) / static_cast<double>(q) + 0.5));
    return {static_cast<int>(a), static_cast<int>(u - a * q)};
}

size_t lds_bytes(const PhaseGeom &g, size_t el) {
    return align16(static_cast<size_t>(g.q + 2 * g.guard) * g.rs * el) +
           static_cast<size_t>(2 * g.n_groups * g.qp) * sizeof(double);
}

}  // namespace

void plan_phase(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *plan, std::vector<int32_t> *table) {
    auto tap_at = [&](int64_t w) -> int { return (w >= -hw && w <= hw) ? tap[w + hw] : 0; };
    double best_cost = 1e300;
    int64_t best_q = 0;
    for (int64_t q = 64; q <= 512; ++q) {
        int64_t nd = 0;
        int guard = 0;
        bool ok = true;
        for (int64_t u = -hw - q; u <= hw && ok; ++u) {
            const bool is_tap = tap_at(u) != 0;
            const bool is_delta = tap_at(u + q) != tap_at(u);
            if (!is_tap && !is_delta) continue;
            nd += is_delta;
            guard = std::max(guard, std::abs(split(u, q).b));
            ok = guard <= kMaxGuard;
        }
        if (!ok || nd == 0 || (nd & 1)) continue;
        const int dp = pad_half(nd / 2);
        if (dp == 0) continue;
        const int64_t qp = (q + 63) / 64 * 64;
        // per output: 2*dp taps at (1 + R)/R vector instructions (R ~ 3) + ~20 for the epilogue
        const double cost = (2.0 * dp * 1.34 + 20.0 + 0.2 * guard) * static_cast<double>(qp) / static_cast<double>(q);
        if (cost < best_cost) {
            best_cost = cost;
            best_q = q;
        }
    }
    plan->phase = PhaseGeom{};
    if (best_q == 0) return;
    const int64_t q = best_q;
    PhaseGeom g{};
    g.q = static_cast<int32_t>(q);
    g.qp = static_cast<int32_t>((q + 63) / 64 * 64);
    std::vector<Split> plus, minus;
    int a_lo = 0, a_hi = 0, guard = 0;
    auto note = [&](const Split &s) {
        a_lo = std::min(a_lo, s.a);
        a_hi = std::max(a_hi, s.a);
        guard = std::max(guard, std::abs(s.b));
    };
    for (int64_t u = -hw - q; u <= hw; ++u) {
        const int d = tap_at(u + q) - tap_at(u);
        if (d == 0) continue;
        const Split s = split(u, q);
        note(s);
        (d > 0 ? plus : minus).push_back(s);
    }
    // full-tap runs in (a, b) space for the per-stretch initialisation
    std::vector<int32_t> runs;  // (a, b_lo, b_hi), a made relative to a_lo below
    for (int64_t u = -hw; u <= hw;) {
        if (!tap_at(u)) {
            ++u;
            continue;
        }
        const Split s = split(u, q);
        note(s);
        int64_t v = u;
        while (v + 1 <= hw && tap_at(v + 1) && split(v + 1, q).a == s.a) ++v;
        note(split(v, q));
        runs.push_back(s.a);
        runs.push_back(s.b);
        runs.push_back(split(v, q).b);
        u = v + 1;
    }
    g.guard = std::max(guard, 1);
    g.d_pad = pad_half(static_cast<int64_t>(plus.size()));
    g.n_runs = static_cast<int32_t>(runs.size() / 3);
    // rows kept behind the current one: the taps' reach, but at least one iteration
    static const int kShapes[][2] = {{2, 4}, {2, 3}, {3, 2}, {2, 2}, {1, 4}, {1, 3}, {1, 2}};
    bool placed = false;
    for (int pass = 0; pass < 2 && !placed; ++pass) {
        for (const auto &sh : kShapes) {
            g.n_groups = sh[0];
            g.rows = sh[1];
            const int ngr = g.n_groups * g.rows;
            g.a_lo = std::min(a_lo, 0);
            // one extra row behind the taps' reach: residues r - b < 0 of the oldest row a tap can
            // touch are mirrored from the row before it, which therefore has to be resident too
            g.a_hi = std::max(a_hi, ngr) + 1;
            g.m_slots = 2 * ngr + (g.a_hi - g.a_lo) + 2;
            g.rs = g.m_slots + g.rows - 1;
            if ((g.rs & 1) == 0) ++g.rs;
            if (g.n_groups * g.qp > 1024) continue;
            if (lds_bytes(g, sizeof(double)) <= (pass == 0 ? kLdsTwoBlocks : kLdsOneBlock)) {
                placed = true;
                break;
            }
        }
    }
    if (!placed) return;
    table->clear();
    auto push_list = [&](const std::vector<Split> &v) {
        for (int t = 0; t < g.d_pad; ++t) table->push_back((t < (int)v.size() ? v[t] : plus[0]).a - g.a_lo);
        for (int t = 0; t < g.d_pad; ++t) table->push_back((t < (int)v.size() ? v[t] : plus[0]).b);
    };
    // padding: the same real tap appended to both lists cancels (n_plus == n_minus always, because
    // shifting the tap set preserves its size)
    push_list(plus);
    push_list(minus);
    for (size_t k = 0; k < runs.size(); k += 3) {
        table->push_back(runs[k] - g.a_lo);
        table->push_back(runs[k + 1]);
        table->push_back(runs[k + 2]);
    }
    plan->phase = g;
}

template <typename TI, typename TO, int DP>
static int launch_phase_r(const parrm_filter_plan *plan, FilterArgs a, PhaseGeom g, hipStream_t stream) {
    void (*kern)(FilterArgs, PhaseGeom) = nullptr;
    switch (g.rows) {
        case 2: kern = filter_phase_kernel<TI, TO, DP, 2>; break;
        case 3: kern = filter_phase_kernel<TI, TO, DP, 3>; break;
        case 4: kern = filter_phase_kernel<TI, TO, DP, 4>; break;
        default: parrm::set_error("filter: unsupported rows-per-thread %d", g.rows); return PARRM_ERR_INVALID;
    }
    const int ngr = g.n_groups * g.rows;
    // stretch: ~128K samples in whole iterations; shrink while the grid would not fill the chip
    int64_t target = 131072;
    if (const char *env = getenv("PARRM_STRETCH_SAMPLES")) target = std::max<int64_t>(atoll(env), g.q);  // tuning knob
    int64_t rows = std::max<int64_t>(ngr, (target / g.q) / ngr * ngr);
    auto blocks_for = [&](int64_t r) { return a.n_chans * ((a.out_len + r * g.q - 1) / (r * g.q)); };
    while (rows > 4 * ngr && blocks_for(rows) < 2048) rows = std::max<int64_t>(ngr, (rows / 2) / ngr * ngr);
    a.stretch_len = rows * g.q;
    a.n_stretch = (a.out_len + a.stretch_len - 1) / a.stretch_len;
    const int64_t blocks = a.n_chans * a.n_stretch;
    PARRM_REQUIRE(blocks <= 0x7fffffffLL, "filter: too many workgroups for one launch");
    const size_t lds = lds_bytes(g, sizeof(TI));
    PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(g.n_groups * g.qp), lds, stream, a, g);
    PARRM_HIP_CHECK(hipGetLastError());
    (void)plan;
    return PARRM_OK;
}

template <typename TI, typename TO>
int launch_phase(const parrm_filter_plan *plan, FilterArgs a, hipStream_t stream) {
    PhaseGeom g = plan->phase;
    g.tab = plan->d_phase_tab;
    if (const char *env = getenv("PARRM_DEBUG_FLAGS")) g.debug = atoi(env);  // profiling ablations only
    switch (g.d_pad) {
        case 4: return launch_phase_r<TI, TO, 4>(plan, a, g, stream);
        case 8: return launch_phase_r<TI, TO, 8>(plan, a, g, stream);
        case 12: return launch_phase_r<TI, TO, 12>(plan, a, g, stream);
        case 14: return launch_phase_r<TI, TO, 14>(plan, a, g, stream);
        case 20: return launch_phase_r<TI, TO, 20>(plan, a, g, stream);
        case 16: return launch_phase_r<TI, TO, 16>(plan, a, g, stream);
        case 24: return launch_phase_r<TI, TO, 24>(plan, a, g, stream);
        case 32: return launch_phase_r<TI, TO, 32>(plan, a, g, stream);
        default: parrm::set_error("filter: no phase kernel for %d delta taps per sign", g.d_pad); return PARRM_ERR_INVALID;
    }
}

template int launch_phase<double, double>(const parrm_filter_plan *, FilterArgs, hipStream_t);
template int launch_phase<float, double>(const parrm_filter_plan *, FilterArgs, hipStream_t);
template int launch_phase<float, float>(const parrm_filter_plan *, FilterArgs, hipStream_t);

}  // namespace parrm_filter
