// filter_data, phase-major kernel with THREE ADJACENT RESIDUES PER LANE ("phase3").
//
// Same recurrence, same residue-major LDS ring and the same workgroup choreography as
// filter_phase_kernel (parrm_filter_phase_impl.h: S(n+q) = S(n) + sum_u d_q(u) xz[n-u], rows of q samples,
// row groups that exchange one partial sum per residue and iteration), but lane j of a row group owns
// residues 3j, 3j+1, 3j+2 of R = 2 consecutive rows instead of one residue:
//
//   * The delta taps of a comb filter are (a) single taps where a tooth edge steps by one residue between
//     two rows -- always a +1 and a -1 in the same row -- and (b) the whole entering and the whole leaving
//     tooth: w adjacent residues of one row.  For (b) the three outputs of a lane need w + 2 adjacent
//     columns instead of 3 w: 9 reads instead of 21 for the 7-wide teeth of the 22 kHz / 130 Hz filter,
//     20 LDS reads per output instead of 28 overall.  LDS bytes are the resource this stencil is closest
//     to, and an exact evaluation cannot read fewer distinct values.
//   * One address per tap serves 3 columns x 2 rows through immediate offsets (the column stride RS is a
//     template constant for that), so address arithmetic and tap-table traffic per output fall ~3x.
//   * Lane stride in LDS is 3*RS elements, RS odd -> still one bank pair per lane for ds_read_b64.
//   * A row group is ONE wave for q + 2B <= 192 (three waves before): 8 waves per CU.  The memory side does
//     not need more (scripts/membench3.hip: 7.39 ms at 8 waves x 6 rows vs 7.38 ms at 24 waves x 2 rows).
//
// Global loads stay contiguous (lane l fills columns l, l + L, l + 2L of a row, L = lanes per group); the
// outputs of a lane are 3 adjacent doubles per row, stored with a 24-byte lane stride.
//
// STATUS (round 2): correct (the whole filter parity suite passes with it), but SLOWER than
// filter_phase_kernel on MI355X -- 10.9-12.1 ms against 9.5-9.7 ms on 256 ch x 10 M f64 -- and therefore
// opt-in (PARRM_PHASE3=1 when the plan is built).  It does what it was built for (LDS instructions -19 %,
// LDS array cycles -16 %, SQ_WAIT_INST_LDS -83 %; gpurun_out/pmc_p3/summary.txt vs
// profiles/r01d_filter_sq_counters.txt), but the vector ALU is as loaded as before (62 instructions per
// output in both kernels: the 28 f64 adds per output are the same sums, and at 4 cycles each they keep a
// SIMD busy 4.6 ms of the launch), and with 8 waves per CU instead of 24 the three co-bottlenecks -- HBM
// stream 8 ms, LDS 5 ms, VALU 4.6 ms -- overlap worse: the tap phase alone (no loads, no stores) takes
// 8.8 ms, i.e. it runs latency-bound at two waves per SIMD.  The ring of the filter's reach (29 rows x 181
// columns x 8 B = 42 KB + the rows in flight) caps a CU at two workgroups whatever the lane mapping.
//
// Guarded form only (comb filters, artefact period >= ~64 samples); everything else stays with
// filter_phase_kernel / filter_stride_kernel.  Compiled once per (input, output) type pair like the
// phase kernel; PARRM_PHASE3_WITH_PLAN adds the host-side planner to one translation unit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "parrm_filter_internal.h"

namespace parrm_filter {

namespace {

typedef unsigned int p3_u32x2_t __attribute__((ext_vector_type(2)));
constexpr int kP3Rows = 2;            // R
constexpr size_t kP3LdsTwoBlocks = 80 * 1024;
constexpr size_t kP3LdsOneBlock = 160 * 1024;

__host__ __device__ inline size_t p3_align16(size_t v) { return (v + 15) & ~size_t{15}; }

__device__ inline unsigned p3_lds_offset(const void *p) {
    return static_cast<unsigned>(reinterpret_cast<size_t>((const __attribute__((address_space(3))) void *)p));
}

// two rows (adjacent elements) of one column: full-rate ds_read_b64 / ds_read_b32 with immediate offsets
// (hipcc would pair them into a half-rate ds_read2; see parrm_filter_phase_impl.h)
template <typename TI, int OFF>
__device__ inline void p3_read_rows(unsigned ad, TI &r0, TI &r1) {
    if constexpr (sizeof(TI) == 8)
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                     : "=&v"(r0), "=&v"(r1) : "v"(ad), "i"(OFF), "i"(OFF + 8));
    else
        asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4"
                     : "=&v"(r0), "=&v"(r1) : "v"(ad), "i"(OFF), "i"(OFF + 4));
}

template <typename TI, int CS, int... K>
__device__ inline void p3_read_cols(unsigned ad, TI (&v)[sizeof...(K)][kP3Rows], std::integer_sequence<int, K...>) {
    (p3_read_rows<TI, K * CS>(ad, v[K][0], v[K][1]), ...);
}

// make values written by the asm reads opaque AFTER a wait statement: their consumers cannot be
// scheduled above it (the compiler does not know the reads are asynchronous)
template <typename TI, int N>
__device__ inline void p3_settle(TI (&v)[N][kP3Rows]) {
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k][0]), "+v"(v[k][1]));
}

// one wide unit: W adjacent taps of one row and sign; residue i of the lane sums columns i .. i+W-1
template <typename TI, int W, int CS>
__device__ inline void p3_wide_unit(unsigned ad, double sgn, double (&delta)[3][kP3Rows]) {
    TI v[W + 2][kP3Rows];
    p3_read_cols<TI, CS>(ad, v, std::make_integer_sequence<int, W + 2>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    p3_settle<TI, W + 2>(v);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < kP3Rows; ++r) {
        // shared middle columns 2 .. W-1 once, then the three windows
        double mid = 0.0;
#pragma unroll
        for (int k = 2; k < W; ++k) mid += static_cast<double>(v[k][r]);
        double s0, s1, s2;
        if constexpr (W == 1) {
            s0 = static_cast<double>(v[0][r]);
            s1 = static_cast<double>(v[1][r]);
            s2 = static_cast<double>(v[2][r]);
        } else {
            s0 = (static_cast<double>(v[0][r]) + static_cast<double>(v[1][r])) + mid;
            s1 = (static_cast<double>(v[1][r]) + static_cast<double>(v[W][r])) + mid;
            s2 = (static_cast<double>(v[W][r]) + static_cast<double>(v[W + 1][r])) + mid;
        }
        delta[0][r] = fma(sgn, s0, delta[0][r]);
        delta[1][r] = fma(sgn, s1, delta[1][r]);
        delta[2][r] = fma(sgn, s2, delta[2][r]);
    }
}

// MAXT: the largest workgroup the instantiation is launched with (one-wave row groups: 4 x 64 threads;
// wider rows: up to 4 x 192) -- the register budget follows from it
template <typename TI, typename TO, int DP1, int RS, int MAXT>
__global__ void __launch_bounds__(MAXT) filter_phase3_kernel(FilterArgs a, Phase3Geom p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int R = kP3Rows;
    constexpr int kEl = static_cast<int>(sizeof(TI));
    constexpr int kElLog2 = kEl == 8 ? 3 : 2;
    constexpr int CS = RS * kEl;  // column stride in bytes
    const int q = p.q, B = p.guard, M = p.m_slots, NG = p.n_groups, L = p.lanes, n_res = p.n_res;
    const int a_lo = p.a_lo, a_hi = p.a_hi;
    const int NGR = NG * R;
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid / L);  // whole waves per group
    const int j = tid - g * L;                               // lane within the group
    const int n_own = (q + 2) / 3;                           // lanes that own at least one residue
    const int j_eff = j < n_own ? j : n_own - 1;             // idle lanes mirror the last owner (broadcast reads)
    const int r0 = 3 * j;                                    // first residue of this lane
    double *xchg = reinterpret_cast<double *>(lds_raw + p3_align16(static_cast<size_t>(n_res + 2) * RS * kEl));
    const int32_t *runs = p.tab + M * p.row_len;

    const int64_t blk = blockIdx.x;
    const int64_t c = blk / a.n_stretch;
    const int64_t st = blk - c * a.n_stretch;
    const int64_t s0 = a.out_first + st * a.stretch_len;
    int64_t s1 = s0 + a.stretch_len;
    if (s1 > a.out_first + a.out_len) s1 = a.out_first + a.out_len;
    const int len = static_cast<int>(s1 - s0);
    const int rows_total = (len + q - 1) / q;
    const TI *xrow = static_cast<const TI *>(a.x) + c * a.ldx;
    TO *yrow = static_cast<TO *>(a.y) + c * a.ldy - a.out_first;
    const int64_t ok_lo = a.buf_first > 0 ? a.buf_first : 0;
    const int64_t ok_hi = a.buf_first + a.buf_len < a.n_total ? a.buf_first + a.buf_len : a.n_total;

    // cell (column col, row) holds sample s0 + row*q + col - B; slot(row) = (row + a_hi) mod M
    const int lane_base = (3 * j_eff + B) * CS;  // byte offset of (first residue of the lane, slot 0)
    auto floor_div = [](int64_t x, int64_t d) -> int64_t { return x >= 0 ? x / d : -((-x + d - 1) / d); };
    const int64_t row_lim = int64_t{1} << 28;
    auto clamp_row = [&](int64_t v) -> int { return static_cast<int>(v < -row_lim ? -row_lim : (v > row_lim ? row_lim : v)); };
    const int lrow_lo = clamp_row(-floor_div(-(ok_lo - s0 + B), q));
    const int lrow_hi = clamp_row(floor_div(ok_hi - n_res - s0 + B, q));
    const int erow_lo = clamp_row(-floor_div(-(static_cast<int64_t>(a.hw) - s0), q));
    const int erow_hi = clamp_row(std::min<int64_t>(floor_div(a.n_total - a.hw - q - s0, q), len / q - 1));

    // fill mapping: lane j fills columns j, j + L, j + 2L of a row
    auto load_cell = [&](int row, int col) -> TI {
        return load_padded(a, xrow, s0 + static_cast<int64_t>(row) * q + col - B);
    };
    auto store_cell = [&](int col, int sl, TI v) {  // (+ mirrored head so that slot + 1 never wraps)
        TI *cell = reinterpret_cast<TI *>(lds_raw + static_cast<size_t>(col) * CS + sl * kEl);
        *cell = v;
        if (sl < R - 1) cell[M] = v;
    };

    // prologue: rows [-a_hi, NGR - a_lo) of xz
    {
        const int n_rows0 = NGR - a_lo + a_hi;
        for (int k = g; k < n_rows0; k += NG) {
            TI v[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[t] = (j + t * L < n_res) ? load_cell(k - a_hi, j + t * L) : TI(0);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (j + t * L < n_res) store_cell(j + t * L, k % M, v[t]);
        }
    }
    // Rows are requested TWO iterations before they are published (three register sets in rotation): with
    // one wave per row group the memory side only stays busy through the tap phase if each wave keeps two
    // iterations of loads in flight.  Rows of iterations 1 and 2 are requested now.
    TI pre_a[R][3], pre_b[R][3], pre_c[R][3];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            pre_a[i][t] = (j + t * L < n_res && NGR < rows_total) ? load_cell(NGR - a_lo + g * R + i, j + t * L) : TI(0);
            pre_b[i][t] = (j + t * L < n_res && 2 * NGR < rows_total) ? load_cell(2 * NGR - a_lo + g * R + i, j + t * L) : TI(0);
        }
    __syncthreads();
    // LDS byte offset of this lane's three fill columns; lanes without a column write a spare column
    unsigned fill_off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) fill_off[t] = static_cast<unsigned>(j + t * L < n_res ? j + t * L : n_res + 1) * CS;

    // S at row 0 for the three residues: full evaluation, once per stretch
    double S[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < p.n_runs; ++k) {
        const int ra = runs[3 * k], b_lo = runs[3 * k + 1], b_hi = runs[3 * k + 2];
        const int sl = (a_hi - a_lo - ra) % M;
        const unsigned char *base = lds_raw + lane_base + sl * kEl;
        for (int b = b_lo; b <= b_hi; ++b) {
#pragma unroll
            for (int i = 0; i < 3; ++i) S[i] += static_cast<double>(*reinterpret_cast<const TI *>(base + (i - b) * CS));
        }
    }

    int sl_top = (g * R - a_lo + a_hi) % M;
    int sl_own = (g * R + a_hi) % M;
    int sl_fill = (NGR - a_lo + g * R + a_hi) % M;
    int par = 0;
    const unsigned lds0 = p3_lds_offset(lds_raw) + static_cast<unsigned>(lane_base);
    auto advance_slots = [&]() {
        par ^= 1;
        sl_top += NGR;
        if (sl_top >= M) sl_top -= M;
        sl_own += NGR;
        if (sl_own >= M) sl_own -= M;
        sl_fill += NGR;
        if (sl_fill >= M) sl_fill -= M;
    };

    unsigned wide_code = 0;
    for (int w = 0; w < p.n_wide; ++w)
        wide_code |= (static_cast<unsigned>(p.wide_w[w] - 1) | (p.wide_sign[w] < 0 ? 8u : 0u)) << (4 * w);
    wide_code = __builtin_amdgcn_readfirstlane(wide_code);
    const int n_wide = (p.debug & 2) ? 0 : p.n_wide;
    typedef const __attribute__((address_space(4))) int32_t *const_i32_ptr;
    int e[2 * DP1 + kP3MaxWide];
    auto fetch_table = [&](int sl) {
        const_i32_ptr trow = (const_i32_ptr)(p.tab) + __builtin_amdgcn_readfirstlane(sl) * p.row_len;
#pragma unroll
        for (int t = 0; t < 2 * DP1; ++t) e[t] = trow[t];
#pragma unroll
        for (int t = 0; t < kP3MaxWide; ++t) e[2 * DP1 + t] = t < p.n_wide ? trow[2 * DP1 + t] : 0;
    };
    fetch_table(sl_top);

    // row sums of the delta taps: delta[i][r] for residue i, row r
    auto tap_sums = [&](double (&delta)[3][R]) {
        double accp[3][R], accm[3][R];
        // single taps in (+1, -1) pairs: 12 reads per pair in flight behind the pair being accumulated
        TI v[2][6][R];  // [register set][+: columns 0..2, -: columns 3..5][row]
        auto issue = [&](int u, TI (&b)[6][R]) {
            const unsigned ap = lds0 + (static_cast<unsigned>(e[u]) << kElLog2);
            const unsigned am = lds0 + (static_cast<unsigned>(e[DP1 + u]) << kElLog2);
            p3_read_rows<TI, 0>(ap, b[0][0], b[0][1]);
            p3_read_rows<TI, CS>(ap, b[1][0], b[1][1]);
            p3_read_rows<TI, 2 * CS>(ap, b[2][0], b[2][1]);
            p3_read_rows<TI, 0>(am, b[3][0], b[3][1]);
            p3_read_rows<TI, CS>(am, b[4][0], b[4][1]);
            p3_read_rows<TI, 2 * CS>(am, b[5][0], b[5][1]);
        };
        auto accumulate = [&](int u, TI (&b)[6][R]) {
            p3_settle<TI, 6>(b);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (u == 0) {
                        accp[i][r] = static_cast<double>(b[i][r]);
                        accm[i][r] = static_cast<double>(b[3 + i][r]);
                    } else {
                        accp[i][r] += static_cast<double>(b[i][r]);
                        accm[i][r] += static_cast<double>(b[3 + i][r]);
                    }
                }
        };
        if (p.debug & 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) delta[i][r] = 0.0;
        } else {
        issue(0, v[0]);
#pragma unroll
        for (int u = 1; u < DP1; ++u) {
            issue(u, v[u & 1]);
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");  // pair u-1 is back (LDS returns in order)
            __builtin_amdgcn_sched_barrier(0);
            accumulate(u - 1, v[(u - 1) & 1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        accumulate(DP1 - 1, v[(DP1 - 1) & 1]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) delta[i][r] = accp[i][r] - accm[i][r];
        }
        // wide units (whole teeth entering / leaving): wave-uniform width dispatch; width - 1 and sign of
        // unit w sit in bits [4w, 4w+3) and bit 4w+3 of one scalar
        for (int w = 0; w < n_wide; ++w) {
            const unsigned ad = lds0 + (static_cast<unsigned>(e[2 * DP1 + w]) << kElLog2);
            const unsigned code = (wide_code >> (4 * w)) & 15u;
            const double sgn = (code & 8u) ? -1.0 : 1.0;
            switch ((code & 7u) + 1) {
                case 1: p3_wide_unit<TI, 1, CS>(ad, sgn, delta); break;
                case 2: p3_wide_unit<TI, 2, CS>(ad, sgn, delta); break;
                case 3: p3_wide_unit<TI, 3, CS>(ad, sgn, delta); break;
                case 4: p3_wide_unit<TI, 4, CS>(ad, sgn, delta); break;
                case 5: p3_wide_unit<TI, 5, CS>(ad, sgn, delta); break;
                case 6: p3_wide_unit<TI, 6, CS>(ad, sgn, delta); break;
                case 7: p3_wide_unit<TI, 7, CS>(ad, sgn, delta); break;
                default: p3_wide_unit<TI, 8, CS>(ad, sgn, delta); break;
            }
        }
    };
    auto own_samples = [&](TI (&xo)[3][R]) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r)
                xo[i][r] = *reinterpret_cast<const TI *>(lds_raw + lane_base + i * CS + (sl_own + r) * kEl);
    };
    // chain the groups after the barrier: S at this wave's first row, then S moves to the next iteration
    const double w0 = g > 0 ? 1.0 : 0.0, w1 = g > 1 ? 1.0 : 0.0, w2 = g > 2 ? 1.0 : 0.0;
    const double u1 = NG > 1 ? 1.0 : 0.0, u2 = NG > 2 ? 1.0 : 0.0, u3 = NG > 3 ? 1.0 : 0.0;
    const int xg1 = (NG > 1 ? 1 : NG - 1) * 3 * L, xg2 = (NG > 2 ? 2 : NG - 1) * 3 * L, xg3 = (NG > 3 ? 3 : NG - 1) * 3 * L;
    auto chain = [&](double (&s_row)[3]) {
        const double *xc = xchg + (par * NG) * 3 * L + j;
        double t[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t[i][0] = xc[i * L];
            t[i][1] = xc[i * L + xg1];
            t[i][2] = xc[i * L + xg2];
            t[i][3] = xc[i * L + xg3];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            s_row[i] = fma(w2, t[i][2], fma(w1, t[i][1], fma(w0, t[i][0], S[i])));
            S[i] += fma(u3, t[i][3], fma(u2, t[i][2], fma(u1, t[i][1], t[i][0])));
        }
    };
    auto next_top = [&]() -> int {
        const int nt = sl_top + NGR;
        return nt >= M ? nt - M : nt;
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // interior iterations go through buffer descriptors (wave-uniform scalar offsets carry the row, one
    // constant per-lane offset per column; lanes without the column get an out-of-range offset)
    constexpr unsigned kNoColumn = 0x80000000u;
    const int64_t xbytes = a.buf_len * kEl, ybytes = a.out_len * static_cast<int64_t>(sizeof(TO));
    const bool fast_ok = !(p.debug & 8) && xbytes < 0x7fffff00LL && ybytes < 0x7fffff00LL;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TI *>(xrow), 0, static_cast<int>(fast_ok ? xbytes : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        yrow + a.out_first, 0, static_cast<int>(fast_ok ? ybytes : 0), 0x00020000);
    unsigned voff_x[3], voff_y[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        voff_x[t] = (j + t * L < n_res) ? static_cast<unsigned>(j + t * L) * kEl : kNoColumn;
        voff_y[t] = (j < n_own && r0 + t < q) ? static_cast<unsigned>(r0 + t) * static_cast<unsigned>(sizeof(TO)) : kNoColumn;
    }
    const unsigned qx = static_cast<unsigned>(q) * kEl, qy = static_cast<unsigned>(q) * static_cast<unsigned>(sizeof(TO));
    // first base row mk of an interior TRIPLE of iterations: every row requested (up to six iterations
    // ahead of mk), published and written by those three iterations is addressable / inside the stretch
    const int pair_lo = std::max(erow_lo, lrow_lo - 3 * NGR + a_lo);
    const int pair_hi = std::min(lrow_hi - 6 * NGR + a_lo + 1, erow_hi - 3 * NGR + 1);
    unsigned soff_x = 0, soff_y = 0;

    auto iteration = [&](const int mk, TI (&pub)[R][3], TI (&req)[R][3], auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        const int m = mk + g * R;
        const bool more = mk + NGR < rows_total;
        const bool more3 = mk + 3 * NGR < rows_total;
        const int frow = mk + NGR - a_lo + g * R;  // first row this wave publishes now
        // 1. request this wave's rows of iteration k+3
        if constexpr (INTERIOR) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (p.debug & 16)
                        req[i][t] = TI(0);
                    else if constexpr (kEl == 8)
                        req[i][t] = __builtin_bit_cast(TI, __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, voff_x[t], soff_x + i * qx, 0));
                    else
                        req[i][t] = __builtin_bit_cast(TI, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, voff_x[t], soff_x + i * qx, 0));
                }
            soff_x += static_cast<unsigned>(NGR) * qx;
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    req[i][t] = (more3 && j + t * L < n_res) ? load_cell(frow + 2 * NGR + i, j + t * L) : TI(0);
        }
        // 2. tap sums, own samples, partial sum for the other groups
        double delta[3][R];
        tap_sums(delta);
        TI xo[3][R];
        own_samples(xo);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double tot = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) tot += delta[i][r];
            xchg[((par * NG + g) * 3 + i) * L + j] = tot;
        }
        // 3. publish the rows requested one iteration ago
        if (INTERIOR || more) {
            if (sl_fill >= R - 1 && sl_fill + R <= M) {  // no wrap, no mirrored head: R adjacent cells per column
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    TI *cell = reinterpret_cast<TI *>(lds_raw + fill_off[t] + sl_fill * kEl);
#pragma unroll
                    for (int i = 0; i < R; ++i) cell[i] = pub[i][t];
                }
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    int sl = sl_fill + i;
                    if (sl >= M) sl -= M;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        TI *cell = reinterpret_cast<TI *>(lds_raw + fill_off[t] + sl * kEl);
                        *cell = pub[i][t];
                        if (sl < R - 1) cell[M] = pub[i][t];
                    }
                }
            }
        }
        lds_barrier();
        double s_row[3];
        chain(s_row);
        fetch_table(next_top());
        // 4. outputs
        if constexpr (INTERIOR) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double yv = fma(-s_row[i], a.inv_taps, static_cast<double>(xo[i][r]));  // (:869 is the repair pass's)
                    const TO yo = static_cast<TO>(yv);
                    if (p.debug & 4) {
                    } else if constexpr (sizeof(TO) == 8)
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(p3_u32x2_t, yo), rsrc_y, voff_y[i], soff_y + r * qy, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yo), rsrc_y, voff_y[i], soff_y + r * qy, 0);
                    s_row[i] += delta[i][r];
                }
            }
            soff_y += static_cast<unsigned>(NGR) * qy;
        } else if (j < n_own) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int rel = (m + r) * q + r0 + i;
                    if (r0 + i < q && rel < len) emit<TO, false>(a, c, s0 + rel, static_cast<double>(xo[i][r]), s_row[i]);
                    s_row[i] += delta[i][r];
                }
        }
        advance_slots();
    };

    int mk = 0;
    while (mk < rows_total) {
        if (fast_ok && mk >= pair_lo && mk <= pair_hi) {
            soff_x = static_cast<unsigned>((s0 - B - a.buf_first + static_cast<int64_t>(mk + 3 * NGR - a_lo + g * R) * q) * kEl);
            soff_y = static_cast<unsigned>((s0 - a.out_first + static_cast<int64_t>(mk + g * R) * q) *
                                           static_cast<int64_t>(sizeof(TO)));
            do {
                iteration(mk, pre_a, pre_c, std::true_type{});
                iteration(mk + NGR, pre_b, pre_a, std::true_type{});
                iteration(mk + 2 * NGR, pre_c, pre_b, std::true_type{});
                mk += 3 * NGR;
            } while (mk <= pair_hi);
        } else {
            iteration(mk, pre_a, pre_c, std::false_type{});
            if (mk + NGR < rows_total) iteration(mk + NGR, pre_b, pre_a, std::false_type{});
            if (mk + 2 * NGR < rows_total) iteration(mk + 2 * NGR, pre_c, pre_b, std::false_type{});
            mk += 3 * NGR;
        }
    }
}

inline size_t p3_lds_bytes(const Phase3Geom &g, size_t el) {
    return p3_align16(static_cast<size_t>(g.n_res + 2) * g.rs * el) +
           static_cast<size_t>(2 * g.n_groups * 3 * g.lanes) * sizeof(double);
}

struct P3Split {
    int a, b;
};
inline P3Split p3_split(int64_t u, int64_t q) {  // nearest multiple: b in (-q/2, q/2]
    const int64_t a = static_cast<int64_t>(std::floor(static_cast<double>(u) / static_cast<double>(q) + 0.5));
    return {static_cast<int>(a), static_cast<int>(u - a * q)};
}

}  // namespace

#ifdef PARRM_PHASE3_WITH_PLAN
// Builds the phase3 geometry + table for the stride the guarded phase plan chose.  Returns false when the
// filter does not fit (wrap-form plans, too many units, LDS).
bool plan_phase3(const std::vector<int8_t> &tap, int64_t hw, const PhaseGeom &phase, Phase3Geom *out,
                 std::vector<int32_t> *table) {
    if (phase.n_groups == 0 || phase.wrap) return false;
    // Opt-in (PARRM_PHASE3=1 at plan time): measured slower than filter_phase_kernel on MI355X, see the
    // header comment and DESIGN.md 4.3.
    const char *env = getenv("PARRM_PHASE3");
    if (!env || atoi(env) == 0) return false;
    const int64_t q = phase.q;
    auto tap_at = [&](int64_t w) -> int { return (w >= -hw && w <= hw) ? tap[w + hw] : 0; };
    std::vector<P3Split> plus, minus;
    int a_lo = 0, a_hi = 0, guard = 0;
    auto note = [&](const P3Split &s) {
        a_lo = std::min(a_lo, s.a);
        a_hi = std::max(a_hi, s.a);
        guard = std::max(guard, std::abs(s.b));
    };
    for (int64_t u = -hw - q; u <= hw; ++u) {
        const int d = tap_at(u + q) - tap_at(u);
        if (d == 0) continue;
        const P3Split s = p3_split(u, q);
        note(s);
        (d > 0 ? plus : minus).push_back(s);
    }
    std::vector<int32_t> runs;
    for (int64_t u = -hw; u <= hw;) {
        if (!tap_at(u)) {
            ++u;
            continue;
        }
        const P3Split s = p3_split(u, q);
        note(s);
        int64_t v = u;
        while (v + 1 <= hw && tap_at(v + 1) && p3_split(v + 1, q).a == s.a) ++v;
        note(p3_split(v, q));
        runs.push_back(s.a);
        runs.push_back(s.b);
        runs.push_back(p3_split(v, q).b);
        u = v + 1;
    }
    Phase3Geom g{};
    g.q = static_cast<int32_t>(q);
    g.guard = std::max(guard, 1);
    g.n_res = g.q + 2 * g.guard;
    g.lanes = ((g.n_res + 2) / 3 + 63) / 64 * 64;
    g.a_lo = std::min(a_lo, 0);
    g.a_hi = std::max(a_hi, 0);
    g.n_runs = static_cast<int32_t>(runs.size() / 3);
    // units: maximal runs of adjacent residues in one row and sign -> wide (width >= 2), else single
    struct Unit {
        int a, b_lo, b_hi;
    };
    auto units_of = [](std::vector<P3Split> v) {
        std::sort(v.begin(), v.end(), [](const P3Split &x, const P3Split &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        std::vector<Unit> out;
        for (size_t i = 0; i < v.size();) {
            size_t k = i;
            while (k + 1 < v.size() && v[k + 1].a == v[i].a && v[k + 1].b == v[k].b + 1 &&
                   static_cast<int>(k + 1 - i) < kP3MaxWide)
                ++k;
            out.push_back({v[i].a, v[i].b, v[k].b});
            i = k + 1;
        }
        return out;
    };
    std::vector<Unit> up = units_of(plus), um = units_of(minus);
    std::vector<Unit> sp, sm;           // singles
    std::vector<std::pair<Unit, int>> wide;  // (unit, sign)
    for (const Unit &u : up) (u.b_lo == u.b_hi ? (void)sp.push_back(u) : (void)wide.push_back({u, +1}));
    for (const Unit &u : um) (u.b_lo == u.b_hi ? (void)sm.push_back(u) : (void)wide.push_back({u, -1}));
    // unpaired singles become width-1 wide units
    while (sp.size() > sm.size()) {
        wide.push_back({sp.back(), +1});
        sp.pop_back();
    }
    while (sm.size() > sp.size()) {
        wide.push_back({sm.back(), -1});
        sm.pop_back();
    }
    if (wide.size() > static_cast<size_t>(kP3MaxWide)) return false;
    int dp1 = 0;
    for (int d : {4, 8, 16})
        if (static_cast<int>(sp.size()) <= d) {
            dp1 = d;
            break;
        }
    if (dp1 == 0 || plus.empty()) return false;
    g.n_pairs = dp1;
    g.n_wide = static_cast<int32_t>(wide.size());
    for (int w = 0; w < g.n_wide; ++w) {
        g.wide_w[w] = wide[w].first.b_hi - wide[w].first.b_lo + 1;
        g.wide_sign[w] = wide[w].second;
    }
    g.row_len = 2 * g.n_pairs + g.n_wide;
    // (groups) best first; the ring must fit next to the exchange area
    bool placed = false;
    for (int pass = 0; pass < 2 && !placed; ++pass) {
        for (int ng : {4, 3, 2, 1}) {
            if (const char *env = getenv("PARRM_PHASE3_GROUPS")) ng = std::max(1, std::min(4, atoi(env)));
            g.n_groups = ng;
            g.m_slots = 2 * ng * kP3Rows + (g.a_hi - g.a_lo);
            g.rs = 0;
            for (int rs : {47, 63})
                if (g.m_slots + kP3Rows - 1 <= rs) {
                    g.rs = rs;
                    break;
                }
            if (g.rs == 0 || g.n_groups * g.lanes > 1024) continue;
            if (p3_lds_bytes(g, sizeof(double)) <= (pass == 0 ? kP3LdsTwoBlocks : kP3LdsOneBlock)) {
                placed = true;
                break;
            }
        }
    }
    if (!placed) return false;
    // a pad pair re-reads one real tap with both signs: it cancels exactly
    auto single_at = [&](const std::vector<Unit> &v, int t) -> Unit {
        if (t < static_cast<int>(v.size())) return v[t];
        if (!sp.empty()) return sp[0];
        return Unit{plus[0].a, plus[0].b, plus[0].b};
    };
    table->clear();
    for (int sl = 0; sl < g.m_slots; ++sl) {
        auto offset = [&](int a, int b_hi) {
            int slot = sl - (a - g.a_lo);
            if (slot < 0) slot += g.m_slots;
            return slot - b_hi * g.rs;
        };
        for (int sign = 0; sign < 2; ++sign)
            for (int t = 0; t < g.n_pairs; ++t) {
                const Unit u = single_at(sign == 0 ? sp : sm, t);
                table->push_back(offset(u.a, u.b_hi));
            }
        for (int w = 0; w < g.n_wide; ++w) table->push_back(offset(wide[w].first.a, wide[w].first.b_hi));
    }
    for (size_t k = 0; k < runs.size(); k += 3) {
        table->push_back(runs[k] - g.a_lo);
        table->push_back(runs[k + 1]);
        table->push_back(runs[k + 2]);
    }
    *out = g;
    return true;
}
#endif  // PARRM_PHASE3_WITH_PLAN

template <typename TI, typename TO, int DP1>
static int launch_phase3_rs(FilterArgs *args, Phase3Geom g, hipStream_t stream) {
    FilterArgs &a = *args;
    void (*kern)(FilterArgs, Phase3Geom) = nullptr;
    const bool narrow = g.n_groups * g.lanes <= 256;
    switch (g.rs * 2 + (narrow ? 1 : 0)) {
        case 47 * 2 + 1: kern = filter_phase3_kernel<TI, TO, DP1, 47, 256>; break;
        case 47 * 2: kern = filter_phase3_kernel<TI, TO, DP1, 47, 768>; break;
        case 63 * 2 + 1: kern = filter_phase3_kernel<TI, TO, DP1, 63, 256>; break;
        case 63 * 2: kern = filter_phase3_kernel<TI, TO, DP1, 63, 768>; break;
        default: parrm::set_error("filter: no phase3 kernel for column stride %d", g.rs); return PARRM_ERR_INVALID;
    }
    const int ngr = g.n_groups * kP3Rows;
    int64_t target = 262144;
    if (const char *env = getenv("PARRM_STRETCH_SAMPLES")) target = std::max<int64_t>(atoll(env), g.q);
    int64_t rows = std::max<int64_t>(ngr, (target / g.q) / ngr * ngr);
    auto blocks_for = [&](int64_t r) { return a.plan_chans * ((a.out_len + r * g.q - 1) / (r * g.q)); };
    if (blocks_for(rows) < 4096 && !getenv("PARRM_STRETCH_SAMPLES")) {  // short recordings: see launch_phase_r
        const double prologue = 16384.0;
        const int64_t resident = 512;
        double best = 1e300;
        int64_t best_rows = rows;
        const int64_t max_stretches = std::max<int64_t>(1, a.out_len / (static_cast<int64_t>(4) * ngr * g.q));
        for (int64_t per_chan = 1; per_chan <= std::min<int64_t>(max_stretches, 4096); ++per_chan) {
            const int64_t r = ((a.out_len + per_chan - 1) / per_chan + g.q - 1) / g.q;
            const int64_t rr = (r + ngr - 1) / ngr * ngr;
            const int64_t blocks = blocks_for(rr);
            const int64_t rounds = (blocks + resident - 1) / resident;
            const double cost = static_cast<double>(rounds) * (static_cast<double>(rr * g.q) + prologue);
            if (cost < best * (1.0 - 1e-9)) {
                best = cost;
                best_rows = rr;
            }
            if (blocks > 8192) break;
        }
        rows = best_rows;
    }
    a.stretch_len = rows * g.q;
    a.n_stretch = (a.out_len + a.stretch_len - 1) / a.stretch_len;
    const int64_t blocks = a.n_chans * a.n_stretch;
    PARRM_REQUIRE(blocks <= 0x7fffffffLL, "filter: too many workgroups for one launch");
    const size_t lds = p3_lds_bytes(g, sizeof(TI));
    PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(g.n_groups * g.lanes), lds, stream, a, g);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

template <typename TI, typename TO>
int launch_phase3(const Phase3Geom &geom, const int32_t *d_tab, FilterArgs *a, hipStream_t stream) {
    Phase3Geom g = geom;
    g.tab = d_tab;
    if (const char *env = getenv("PARRM_P3_DEBUG")) g.debug = atoi(env);  // profiling ablations only
    switch (g.n_pairs) {
        case 4: return launch_phase3_rs<TI, TO, 4>(a, g, stream);
        case 8: return launch_phase3_rs<TI, TO, 8>(a, g, stream);
        case 16: return launch_phase3_rs<TI, TO, 16>(a, g, stream);
        default: parrm::set_error("filter: no phase3 kernel for %d single-tap pairs", g.n_pairs); return PARRM_ERR_INVALID;
    }
}

template int launch_phase3<PARRM_PHASE_TI, PARRM_PHASE_TO>(const Phase3Geom &, const int32_t *, FilterArgs *, hipStream_t);

}  // namespace parrm_filter
