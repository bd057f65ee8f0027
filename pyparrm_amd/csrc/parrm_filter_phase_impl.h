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
//
// This file is the implementation; it is compiled three times (parrm_filter_phase.hip, _fd.hip,
// _ff.hip: one translation unit per (input, output) type pair, so that the ~150 kernel instantiations
// build in parallel).  PARRM_PHASE_TI / PARRM_PHASE_TO select the pair, PARRM_PHASE_WITH_PLAN adds
// the host-side planner to exactly one of them.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "parrm_filter_internal.h"

namespace parrm_filter {

namespace {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
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
// back, whatever other LDS traffic the compiler had in flight before the statement.  (Scalar loads
// share lgkmcnt but return out of order: the kernel keeps none in flight across the tap loop --
// the tap-table row is requested behind the barrier and waited for before the first tap.)
// The LDS pipe therefore never drains inside the tap loop (draining every batch measured ~45 %
// of the LDS rate at 3 waves/SIMD).  Uses of a batch are kept below its wait by a
// sched_barrier (the compiler does not know the asm results are asynchronous).
#define PARRM_RD3(op, o0, o1, o2, a, s) \
    op " %" #o0 ", %" #a "\n\t" op " %" #o1 ", %" #a " offset:" #s "\n\t" op " %" #o2 ", %" #a " offset:2*" #s "\n\t"
#define PARRM_RD4(op, o0, o1, o2, o3, a, s)                                                              \
    op " %" #o0 ", %" #a "\n\t" op " %" #o1 ", %" #a " offset:" #s "\n\t" op " %" #o2 ", %" #a " offset:2*" #s \
       "\n\t" op " %" #o3 ", %" #a " offset:3*" #s "\n\t"
// One row-set = R consecutive elements of one tap.  WAIT = n: after issuing, wait until at most n
// LDS operations are outstanding (n = 3*R keeps three row-sets in flight; lgkmcnt is 4 bits).
#define PARRM_WAITSTR(n) "s_waitcnt lgkmcnt(" #n ")"
template <typename TI, int R, int WAIT>
__device__ inline void tap_rows(unsigned ad, TI (&v)[R]) {
    static_assert(WAIT == -1 || WAIT == 9 || WAIT == 12, "wait count");
    if constexpr (sizeof(TI) == 8 && R == 4) {
        if constexpr (WAIT == 12)
            asm volatile(PARRM_RD4("ds_read_b64", 0, 1, 2, 3, 4, 8) PARRM_WAITSTR(12)
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad));
        else
            asm volatile(PARRM_RD4("ds_read_b64", 0, 1, 2, 3, 4, 8) ""
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad));
    } else if constexpr (sizeof(TI) == 4 && R == 4) {
        if constexpr (WAIT == 12)
            asm volatile(PARRM_RD4("ds_read_b32", 0, 1, 2, 3, 4, 4) PARRM_WAITSTR(12)
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad));
        else
            asm volatile(PARRM_RD4("ds_read_b32", 0, 1, 2, 3, 4, 4) ""
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad));
    } else if constexpr (sizeof(TI) == 8 && R == 3) {
        if constexpr (WAIT == 9)
            asm volatile(PARRM_RD3("ds_read_b64", 0, 1, 2, 3, 8) PARRM_WAITSTR(9)
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(ad));
        else
            asm volatile(PARRM_RD3("ds_read_b64", 0, 1, 2, 3, 8) ""
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(ad));
    } else {
        static_assert(R == 3 && sizeof(TI) == 4, "unsupported row-set shape (R == 2 goes through tap_pair)");
        if constexpr (WAIT == 9)
            asm volatile(PARRM_RD3("ds_read_b32", 0, 1, 2, 3, 4) PARRM_WAITSTR(9)
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(ad));
        else
            asm volatile(PARRM_RD3("ds_read_b32", 0, 1, 2, 3, 4) ""
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(ad));
    }
}

// R == 2: one unit = the +1 and the -1 row-set of one delta pair (4 reads, two addresses), so that
// one counted wait covers four reads.  WAIT = 4: on return the previous unit is back.
template <typename TI, int WAIT>
__device__ inline void tap_pair(unsigned ad_p, unsigned ad_m, TI (&v)[4]) {
    static_assert(WAIT == -1 || WAIT == 4, "wait count");
    if constexpr (sizeof(TI) == 8) {
        if constexpr (WAIT == 4)
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8\n\t"
                         "s_waitcnt lgkmcnt(4)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad_p), "v"(ad_m));
        else
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad_p), "v"(ad_m));
    } else {
        if constexpr (WAIT == 4)
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %5\n\tds_read_b32 %3, %5 offset:4\n\t"
                         "s_waitcnt lgkmcnt(4)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad_p), "v"(ad_m));
        else
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %5\n\tds_read_b32 %3, %5 offset:4"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ad_p), "v"(ad_m));
    }
}

// WRAP: the general-residue form for SHORT artefact periods.  There the taps do not sit near the
// multiples of any stride q but at several residues of it, so a delta tap u = a*q + b is split with
// b in [0, q) and lane r reads residue r - b of row m - a when r >= b, and residue r - b + q of row
// m - a - 1 when r < b: the same address plus the constant (q*RS - 1) elements.  Which taps wrap is a
// per-lane bit mask fixed for the kernel; every column carries one extra slot in front (a copy of
// the last slot) so that "one slot earlier" never leaves the column.  No mirrored residues (B = 0).
//
// PACK (float32 recordings, R == 2, guarded form): the two rows of a tap are ONE aligned 8-byte read and
// one packed float32 add.  With 4-byte cells the pair (slot, slot + 1) of a column is 8-byte aligned
// only for even slots (a ds_read_b64 off its alignment is replayed at 64 cycles), so the ring is kept
// twice: copy E holds row slot s at element s, copy O at element s + 1; the plan's table points every
// (table row, tap) at the copy in which its pair is aligned -- a byte offset, no run-time select.  The
// column stride RS is even with RS/2 odd there (lane stride RS*4 B keeps ds_read_b64 conflict-free).
// PACK = 1 (float64 output, the reference's dtype rule): the tap values are still converted and summed in
// float64 -- results unchanged to the bit, half the LDS instructions.  PACK = 2 (float32 output, the
// build option of BASELINE configs[4]): the 14 + 14 tap values of a row are summed in float32
// (v_pk_add_f32: both rows per instruction); the running sum S and everything downstream stay float64.
// The reference's own float32 path is an FFT in float32 (~1e-6 of the sample scale); this keeps ~4e-7.
// MODE 1 (segmented plans, parrm_filter.hip: half-widths beyond any LDS ring; PhaseGeom::debug bit 5: store
// instead of add, the first pass of a call): the launch evaluates one
// SEGMENT of a long filter's taps and ADDS the raw zero-padded tap sum S(n) into the float64 accumulator
// y[n] (global_atomic_add_f64 without return: no read latency in the loop; the passes of one call follow
// each other on the stream and every output is touched once per pass, so the result is deterministic).  No
// division, no own sample, no edge divisors: the caller's combine kernel applies those for the whole filter.
template <typename TI, typename TO, int DP, int R, bool WRAP, int PACK = 0, int MODE = 0>
__global__ void __launch_bounds__(1024) filter_phase_kernel(FilterArgs a, PhaseGeom p) {
    static_assert(PACK == 0 || (sizeof(TI) == 4 && R == 2 && !WRAP), "PACK: float32 input, two rows, guarded form");
    static_assert(MODE == 0 || (sizeof(TO) == 8 && PACK != 2), "MODE 1 accumulates float64 sums");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int kEl = static_cast<int>(sizeof(TI));
    constexpr int kElLog2 = kEl == 8 ? 3 : 2;
    const int q = p.q, QP = p.qp, B = p.guard, M = p.m_slots, RS = p.rs, NG = p.n_groups;
    const int a_lo = p.a_lo, a_hi = p.a_hi;
    const int NGR = NG * R;
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid / QP);  // whole waves per group
    const int r = tid - g * QP;
    const bool active = r < q;            // compute lane: owns residue r
    const bool filler = r < q + 2 * B;    // fill lane: owns cell column rho (idle lanes take the halo)
    const int rho = r < q + B ? r : r - q - 2 * B;  // extended residue in [-B, q+B)
    const int n_res = q + 2 * B;
    const int copy_o = PACK != 0 ? p.copy_o_bytes : 0;  // byte offset of copy O from copy E
    double *xchg = reinterpret_cast<double *>(lds_raw + (PACK != 0 ? 2 : 1) * align16(static_cast<size_t>(n_res) * RS * kEl));
    const int32_t *runs = p.tab + M * 2 * DP;

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
    // addressable, in-recording sample range (everything else reads as zero)
    const int64_t ok_lo = a.buf_first > 0 ? a.buf_first : 0;
    const int64_t ok_hi = a.buf_first + a.buf_len < a.n_total ? a.buf_first + a.buf_len : a.n_total;

    // Cell (rho, row) holds sample s0 + row*q + rho for every rho in [-B, q+B): the B-wide halos are
    // ordinary columns owned by the otherwise idle lanes of the last wave.  slot(row) = (row+a_hi) mod M.
    constexpr int kFront = WRAP ? 1 : 0;                         // slots in front of slot 0
    const int fill_base = ((rho + B) * RS + kFront) * kEl;       // byte offset of (rho, slot 0)
    // (idle lanes mirror the last residue: same address as a lane of their own half -> broadcast, no bank conflict)
    const int r_eff = active ? r : q - 1;
    const int lane_base = ((r_eff + B) * RS + kFront) * kEl;      // byte offset of (r, slot 0)
    // Rows whose whole span (halos included) is addressable take a wave-uniform fast path: scalar
    // base pointer + per-lane unsigned offset (no per-lane bounds tests).  Computed once per stretch.
    auto floor_div = [](int64_t x, int64_t d) -> int64_t { return x >= 0 ? x / d : -((-x + d - 1) / d); };
    const int64_t row_lim = int64_t{1} << 28;
    auto clamp_row = [&](int64_t v) -> int { return static_cast<int>(v < -row_lim ? -row_lim : (v > row_lim ? row_lim : v)); };
    // load: first = s0 + row*q - B >= ok_lo  and  first + n_res <= ok_hi
    const int lrow_lo = clamp_row(-floor_div(-(ok_lo - s0 + B), q));
    const int lrow_hi = clamp_row(floor_div(ok_hi - n_res - s0 + B, q));
    // emit: n0 = s0 + row*q >= hw, n0 + q - 1 + hw < n_total, (row + 1)*q <= len
    const int erow_lo = clamp_row(-floor_div(-(static_cast<int64_t>(a.hw) - s0), q));
    const int erow_hi = clamp_row(std::min<int64_t>(floor_div(a.n_total - a.hw - q - s0, q), len / q - 1));
    const TI *xfast = xrow + (s0 - a.buf_first - B);  // wave-uniform
    const unsigned fill_off = static_cast<unsigned>(rho + B);
    auto load_row = [&](int row) -> TI {  // this lane's sample of `row`
        if (row >= lrow_lo && row <= lrow_hi) return (xfast + static_cast<int64_t>(row) * q)[fill_off];
        return load_padded(a, xrow, s0 + static_cast<int64_t>(row) * q + rho);
    };
    auto store_row = [&](int sl, TI v) {  // (+ mirrored head so that slot + i, i < R, never wraps)
        TI *cell = reinterpret_cast<TI *>(lds_raw + fill_base + sl * kEl);
        *cell = v;
        if (sl < R - 1) cell[M] = v;
        if (WRAP && sl == M - 1) cell[-M] = v;  // the slot in front of slot 0
        if constexpr (PACK != 0) {  // copy O: row slot s at element s + 1 (+ its mirrored head)
            TI *cell_o = reinterpret_cast<TI *>(lds_raw + copy_o + fill_base + (sl + 1) * kEl);
            *cell_o = v;
            if (sl < R - 1 && !(M & 1)) cell_o[M] = v;  // (odd M: that element is never read, see the planner)
        }
    };

    // prologue: rows [-a_hi, NGR - a_lo) of xz, eight loads per lane in flight
    {
        const int n_rows0 = NGR - a_lo + a_hi;
        const int n_grp = blockDim.x / QP;  // == NG
        for (int k0 = 0; k0 < n_rows0; k0 += 8 * n_grp) {
            TI v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * n_grp + g;
                v[j] = (k < n_rows0 && filler) ? load_row(k - a_hi) : TI(0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * n_grp + g;
                if (k < n_rows0 && filler) store_row(k % M, v[j]);
            }
        }
    }
    // rows of iteration 1 are requested now (published during iteration 0)
    TI pre_a[R];
#pragma unroll
    for (int i = 0; i < R; ++i)
        pre_a[i] = (filler && NGR < rows_total) ? load_row(NGR - a_lo + g * R + i) : TI(0);
    __syncthreads();

    // wrap: a tap with residue offset b > r reads (residue r - b + q, one row earlier)
    const int wrap_corr = (q * RS - 1) * kEl;
    unsigned wrap_p = 0, wrap_m = 0;  // bit t: this lane wraps +1 tap t / -1 tap t
    if constexpr (WRAP) {
        const int32_t *bl = runs + 3 * p.n_runs;
        for (int t = 0; t < DP; ++t) {
            wrap_p |= (r_eff < bl[t] ? 1u : 0u) << t;
            wrap_m |= (r_eff < bl[DP + t] ? 1u : 0u) << t;
        }
    }

    // S at row 0 for this residue: full evaluation, once per stretch (every group, redundantly)
    double S = 0.0;
    if (active) {
        double s_a = 0.0, s_b = 0.0;
        for (int k = 0; k < p.n_runs; ++k) {
            const int ra = runs[3 * k], b_lo = runs[3 * k + 1], b_hi = runs[3 * k + 2];
            const int sl = (a_hi - a_lo - ra) % M;  // slot(0 - a), a = ra + a_lo
            const unsigned char *base = lds_raw + lane_base + sl * kEl;
            if constexpr (WRAP) {
                for (int b = b_lo; b <= b_hi; ++b)
                    s_a += static_cast<double>(
                        *reinterpret_cast<const TI *>(base - b * RS * kEl + (r < b ? wrap_corr : 0)));
            } else {
                int b = b_lo;
                for (; b + 1 <= b_hi; b += 2) {
                    s_a += static_cast<double>(*reinterpret_cast<const TI *>(base - b * RS * kEl));
                    s_b += static_cast<double>(*reinterpret_cast<const TI *>(base - (b + 1) * RS * kEl));
                }
                if (b <= b_hi) s_a += static_cast<double>(*reinterpret_cast<const TI *>(base - b * RS * kEl));
            }
        }
        S = s_a + s_b;
    }

    // scalar slot bookkeeping for this wave's first row m = mk + g*R
    int sl_top = (g * R - a_lo + a_hi) % M;         // slot(m - a_lo): the tap table is indexed by it
    int sl_own = (g * R + a_hi) % M;                // slot(m)
    int sl_fill = (NGR - a_lo + g * R + a_hi) % M;  // slot of the first row this wave publishes
    int par = 0;
    const unsigned lds0 = lds_offset(lds_raw) + static_cast<unsigned>(lane_base);
    auto advance_slots = [&]() {
        par ^= 1;
        sl_top += NGR;
        if (sl_top >= M) sl_top -= M;
        sl_own += NGR;
        if (sl_own >= M) sl_own -= M;
        sl_fill += NGR;
        if (sl_fill >= M) sl_fill -= M;
    };

    // Tap offsets for one iteration: one table row, wave-uniform -> scalar loads.  Constant address
    // space + readfirstlane: the table is never written by this kernel, which the compiler cannot
    // prove for a plain global pointer once the kernel has stored to y -- it then fetches the row
    // with per-lane global loads whose vmcnt wait also drains the row prefetch and the previous
    // iteration's stores every iteration.
    typedef const __attribute__((address_space(4))) int32_t *const_i32_ptr;
    // 2. row sums Delta_i = sum_u d_q(u) xz[(m+i)q + r - u], and this lane's own samples of the rows
    int e[2 * DP];
    auto fetch_table = [&](int sl) {
        const_i32_ptr trow = (const_i32_ptr)(p.tab) + __builtin_amdgcn_readfirstlane(sl) * (2 * DP);
#pragma unroll
        for (int t = 0; t < 2 * DP; ++t) e[t] = trow[t];
    };
    fetch_table(sl_top);  // first iteration's row; later rows are requested right after each barrier
    auto tap_sums = [&](double (&delta)[R], const bool skip_taps) {
        double accp[R], accm[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            accp[i] = 0.0;
            accm[i] = 0.0;
        }
        if (skip_taps) {
        } else if constexpr (PACK != 0) {
            // one unit = the +1 and the -1 tap of a pair: two aligned 8-byte reads (both rows each), two
            // packed adds; three register sets keep two units in flight behind the one being added
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            f32x2_t sp = {0.0f, 0.0f}, sm = {0.0f, 0.0f};  // PACK == 2: float32 sums
            f32x2_t vp[3], vm[3];
            auto add = [&](const f32x2_t &bp, const f32x2_t &bm) {
                if constexpr (PACK == 2) {
                    sp += bp;
                    sm += bm;
                } else {  // PACK == 1: float64 sums, as the unpacked kernel forms them
                    accp[0] += static_cast<double>(bp.x);
                    accp[1] += static_cast<double>(bp.y);
                    accm[0] += static_cast<double>(bm.x);
                    accm[1] += static_cast<double>(bm.y);
                }
            };
            auto issue = [&](int u, f32x2_t &bp, f32x2_t &bm, auto wait) {
                const unsigned ap = lds0 + static_cast<unsigned>(e[u]);       // the table holds byte offsets
                const unsigned am = lds0 + static_cast<unsigned>(e[DP + u]);
                if constexpr (decltype(wait)::value == 4)
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(4)"
                                 : "=&v"(bp), "=&v"(bm) : "v"(ap), "v"(am));
                else
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3" : "=&v"(bp), "=&v"(bm) : "v"(ap), "v"(am));
            };
            issue(0, vp[0], vm[0], std::integral_constant<int, -1>{});
            if constexpr (DP > 1) issue(1, vp[1], vm[1], std::integral_constant<int, -1>{});
#pragma unroll
            for (int u = 2; u < DP; ++u) {
                issue(u, vp[u % 3], vm[u % 3], std::integral_constant<int, 4>{});  // returns once unit u-2 is back
                __builtin_amdgcn_sched_barrier(0);
                add(vp[(u - 2) % 3], vm[(u - 2) % 3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DP > 1) add(vp[(DP - 2) % 3], vm[(DP - 2) % 3]);
            add(vp[(DP - 1) % 3], vm[(DP - 1) % 3]);
            if constexpr (PACK == 2) {
                accp[0] = static_cast<double>(sp.x);
                accp[1] = static_cast<double>(sp.y);
                accm[0] = static_cast<double>(sm.x);
                accm[1] = static_cast<double>(sm.y);
            }
        } else if constexpr (R == 2) {
            // DP units of (+1 row-set, -1 row-set); two register sets: one unit in flight behind the
            // one being accumulated, one counted wait per four reads
            TI v[2][4];
            auto issue = [&](int u, TI (&b)[4], auto wait) {
                unsigned ap = lds0 + (static_cast<unsigned>(e[u]) << kElLog2);
                unsigned am = lds0 + (static_cast<unsigned>(e[DP + u]) << kElLog2);
                if constexpr (WRAP) {
                    ap += __builtin_amdgcn_ubfe(wrap_p, u, 1) * static_cast<unsigned>(wrap_corr);
                    am += __builtin_amdgcn_ubfe(wrap_m, u, 1) * static_cast<unsigned>(wrap_corr);
                }
                tap_pair<TI, decltype(wait)::value>(ap, am, b);
            };
            auto accumulate = [&](int u, TI (&b)[4]) {
                if (u == 0) {
                    accp[0] = static_cast<double>(b[0]);
                    accp[1] = static_cast<double>(b[1]);
                    accm[0] = static_cast<double>(b[2]);
                    accm[1] = static_cast<double>(b[3]);
                } else {
                    accp[0] += static_cast<double>(b[0]);
                    accp[1] += static_cast<double>(b[1]);
                    accm[0] += static_cast<double>(b[2]);
                    accm[1] += static_cast<double>(b[3]);
                }
            };
            issue(0, v[0], std::integral_constant<int, -1>{});
#pragma unroll
            for (int u = 1; u < DP; ++u) {
                issue(u, v[u & 1], std::integral_constant<int, 4>{});  // returns once unit u-1 is back
                __builtin_amdgcn_sched_barrier(0);
                accumulate(u - 1, v[(u - 1) & 1]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            accumulate(DP - 1, v[(DP - 1) & 1]);
        } else {
            // 2*DP row-sets, plus and minus interleaved; a ring of four register sets keeps three
            // row-sets (3*R reads) in flight behind the one being accumulated
            constexpr int NS = 2 * DP;
            constexpr int W3 = 3 * R;
            TI v[4][R];
            auto addr = [&](int j) -> unsigned {  // j even: +1 tap j/2, j odd: -1 tap j/2
                unsigned ad = lds0 + (static_cast<unsigned>(e[(j & 1) * DP + (j >> 1)]) << kElLog2);
                if constexpr (WRAP)
                    ad += __builtin_amdgcn_ubfe((j & 1) ? wrap_m : wrap_p, j >> 1, 1) * static_cast<unsigned>(wrap_corr);
                return ad;
            };
            auto accumulate = [&](int j, TI (&b)[R]) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (j == 0)
                        accp[i] = static_cast<double>(b[i]);  // first +1 tap: no zero-init needed
                    else if (j == 1)
                        accm[i] = static_cast<double>(b[i]);  // first -1 tap
                    else if (j & 1)
                        accm[i] += static_cast<double>(b[i]);
                    else
                        accp[i] += static_cast<double>(b[i]);
                }
            };
            tap_rows<TI, R, -1>(addr(0), v[0]);
            tap_rows<TI, R, -1>(addr(1), v[1]);
            tap_rows<TI, R, -1>(addr(2), v[2]);
#pragma unroll
            for (int j = 3; j < NS; ++j) {
                tap_rows<TI, R, W3>(addr(j), v[j & 3]);  // issues row-set j; returns once j-3 is back
                __builtin_amdgcn_sched_barrier(0);
                accumulate(j - 3, v[(j - 3) & 3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            accumulate(NS - 3, v[(NS - 3) & 3]);
            accumulate(NS - 2, v[(NS - 2) & 3]);
            accumulate(NS - 1, v[(NS - 1) & 3]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) delta[i] = accp[i] - accm[i];
    };
    // this lane's own samples of the rows (needed only for the outputs: requested after the tap loop
    // so that nothing has to drain in front of it)
    auto own_samples = [&](TI (&xo)[R]) {
#pragma unroll
        for (int i = 0; i < R; ++i) xo[i] = *reinterpret_cast<const TI *>(lds_raw + lane_base + (sl_own + i) * kEl);
    };
    // 4. chain the groups after the barrier: returns S at this wave's first row and moves S to the
    // next iteration's base row
    // Straight-line for any NG <= 4: all four cells are read at once (groups beyond NG re-read the last
    // real one) and combined with wave-uniform 0/1 weights -- fma(1, t, s) rounds like s + t and
    // fma(0, t, s) is s -- instead of per-group branches, which serialised one LDS round trip each.
    const double w0 = g > 0 ? 1.0 : 0.0, w1 = g > 1 ? 1.0 : 0.0, w2 = g > 2 ? 1.0 : 0.0;
    const double u1 = NG > 1 ? 1.0 : 0.0, u2 = NG > 2 ? 1.0 : 0.0, u3 = NG > 3 ? 1.0 : 0.0;
    const int xg1 = (NG > 1 ? 1 : NG - 1) * QP, xg2 = (NG > 2 ? 2 : NG - 1) * QP, xg3 = (NG > 3 ? 3 : NG - 1) * QP;
    auto chain_reads = [&](double (&t)[4]) {
        const double *xc = xchg + (par * NG) * QP + r;
        t[0] = xc[0];
        t[1] = xc[xg1];
        t[2] = xc[xg2];
        t[3] = xc[xg3];
    };
    auto chain_math = [&](const double (&t)[4]) -> double {
        const double s_row = fma(w2, t[2], fma(w1, t[1], fma(w0, t[0], S)));
        S += fma(u3, t[3], fma(u2, t[2], fma(u1, t[1], t[0])));
        return s_row;
    };
    auto next_top = [&]() -> int {
        const int nt = sl_top + NGR;
        return nt >= M ? nt - M : nt;
    };
    // LDS-only barrier: a plain __syncthreads() also waits vmcnt(0), i.e. for the global loads
    // requested above and for the previous iteration's output stores, once per iteration
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // One iteration = NG*R rows.  `pub` holds the rows requested one iteration ago (published to LDS
    // here), `req` receives the rows of iteration k+2.  The loops below alternate two register sets so
    // that no copy ever touches a register with a load in flight (a copy would force vmcnt(0)).
    //
    // General form: any iteration of the stretch (recording edges, stretch tail, ablation flags).
    auto iteration = [&](const int mk, TI (&pub)[R], TI (&req)[R]) {
        const int m = mk + g * R;
        const bool more = mk + NGR < rows_total;
        const bool more2 = mk + 2 * NGR < rows_total;
        // 1. request this lane's share of the rows iteration k+2 needs.  One wave-uniform test
        // covers the R rows; the interior case is R loads off a wave-uniform base pointer.
        const int frow = mk + NGR - a_lo + g * R;  // first row this wave publishes now
        if (more2 && !(p.debug & 4)) {
            if (frow + NGR >= lrow_lo && frow + NGR + R - 1 <= lrow_hi) {
                const TI *xreq = xfast + static_cast<int64_t>(frow + NGR) * q;
#pragma unroll
                for (int i = 0; i < R; ++i) req[i] = filler ? (xreq + static_cast<int64_t>(i) * q)[fill_off] : TI(0);
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) req[i] = filler ? load_row(frow + NGR + i) : TI(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) req[i] = TI(0);
        }
        TI xo[R];
        double delta[R];
        tap_sums(delta, (p.debug & 1) != 0);
        if constexpr (MODE == 0) own_samples(xo);
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) tot += delta[i];
        xchg[(par * NG + g) * QP + r] = tot;
        // 3. publish the rows requested one iteration ago (their slots alias rows older than mk - a_hi)
        if (more && filler && !(p.debug & 4)) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int sl = sl_fill + i;
                if (sl >= M) sl -= M;
                store_row(sl, pub[i]);
            }
        }
        if (!(p.debug & 8)) lds_barrier();
        double tg[4];
        chain_reads(tg);
        fetch_table(next_top());  // next iteration's tap offsets: one wait covers them and the cells
        double s_row = chain_math(tg);
        // 5. outputs.  Rows that are interior to the recording and to the stretch take a branch-free
        // path (the test is wave-uniform); edge rows go through emit().
        if constexpr (MODE == 1) {
            if (active) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int rel = (m + i) * q + r;
                    if (rel < len) {
                        double *cell = reinterpret_cast<double *>(yrow) + (s0 + rel);
                        if (p.debug & 32)
                            *cell = s_row;  // first pass of a call
                        else
                            __builtin_amdgcn_global_atomic_fadd_f64(cell, s_row);
                    }
                    s_row += delta[i];
                }
            }
        } else if (!(p.debug & 2)) {
            if (m >= erow_lo && m + R - 1 <= erow_hi) {
                if (active) {
                    TO *yout = yrow + (s0 + static_cast<int64_t>(m) * q);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const double yv = fma(-s_row, a.inv_taps, static_cast<double>(xo[i]));  // (:869 is the repair pass's)
                        (yout + static_cast<int64_t>(i) * q)[static_cast<unsigned>(r)] = static_cast<TO>(yv);
                        s_row += delta[i];
                    }
                }
            } else if (active) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int rel = (m + i) * q + r;
                    if (rel < len) emit<TO, false>(a, c, s0 + rel, static_cast<double>(xo[i]), s_row);
                    s_row += delta[i];
                }
            }
        }
        advance_slots();
    };

    // Interior form: every row requested, published and written by the NEXT TWO iterations of every
    // wave of the workgroup lies inside the recording and the stretch, and byte offsets fit 31 bits.
    // Rows go through buffer descriptors: wave-uniform scalar offsets carry the row position, one
    // constant per-lane offset carries the column, and lanes without a column get an out-of-range
    // offset (the range check returns 0 for their loads and drops their stores) -- no per-lane
    // address arithmetic, no bounds tests, no exec juggling.
    constexpr unsigned kNoColumn = 0x80000000u;
    const int64_t xbytes = a.buf_len * kEl, ybytes = a.out_len * static_cast<int64_t>(sizeof(TO));
    const bool fast_ok = !(p.debug & 31) && xbytes < 0x7fffff00LL && ybytes < 0x7fffff00LL;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TI *>(xrow), 0, static_cast<int>(fast_ok ? xbytes : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        yrow + a.out_first, 0, static_cast<int>(fast_ok ? ybytes : 0), 0x00020000);
    const unsigned voff_x = filler ? fill_off * kEl : kNoColumn;
    const unsigned voff_y = active ? static_cast<unsigned>(r) * static_cast<unsigned>(sizeof(TO)) : kNoColumn;
    const unsigned qx = static_cast<unsigned>(q) * kEl, qy = static_cast<unsigned>(q) * static_cast<unsigned>(sizeof(TO));
    // first base row mk of an interior PAIR of iterations: mk in [pair_lo, pair_hi]
    const int pair_lo = std::max(erow_lo, lrow_lo - 2 * NGR + a_lo);
    // (no "more rows to request in this stretch" condition: near the stretch tail the requests run
    // into the next stretch's samples, which are addressable and simply never used)
    const int pair_hi = std::min(lrow_hi - 4 * NGR + a_lo + 1, erow_hi - 2 * NGR + 1);
    unsigned soff_x = 0, soff_y = 0;  // set on entry to the interior loop
    auto fast_iteration = [&](TI (&pub)[R], TI (&req)[R]) {
        // 1. request rows (iteration k+2)
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if constexpr (kEl == 8) {
                req[i] = __builtin_bit_cast(TI, __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, voff_x, soff_x + i * qx, 0));
            } else {
                req[i] = __builtin_bit_cast(TI, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, voff_x, soff_x + i * qx, 0));
            }
        }
        soff_x += static_cast<unsigned>(NGR) * qx;
        TI xo[R];
        double delta[R];
        tap_sums(delta, false);
        if constexpr (MODE == 0) own_samples(xo);
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) tot += delta[i];
        xchg[(par * NG + g) * QP + r] = tot;
        // 3. publish
        if (filler) {
            // no wrap, no mirrored head (nor, in the wrap form, the mirrored last slot): R adjacent cells
            if (sl_fill >= R - 1 && sl_fill + R <= M - kFront) {
                TI *cell = reinterpret_cast<TI *>(lds_raw + fill_base + sl_fill * kEl);
#pragma unroll
                for (int i = 0; i < R; ++i) cell[i] = pub[i];
                if constexpr (PACK != 0) {
                    TI *cell_o = reinterpret_cast<TI *>(lds_raw + copy_o + fill_base + (sl_fill + 1) * kEl);
#pragma unroll
                    for (int i = 0; i < R; ++i) cell_o[i] = pub[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    int sl = sl_fill + i;
                    if (sl >= M) sl -= M;
                    store_row(sl, pub[i]);
                }
            }
        }
        lds_barrier();
        double tg[4];
        chain_reads(tg);
        fetch_table(next_top());  // next iteration's tap offsets: one wait covers them and the cells
        double s_row = chain_math(tg);
        // 5. outputs
        if constexpr (MODE == 1) {
            if (active) {
                double *acc = reinterpret_cast<double *>(yrow + a.out_first) + (soff_y >> 3) + r;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (p.debug & 32)
                        acc[static_cast<int64_t>(i) * q] = s_row;  // first pass of a call
                    else
                        __builtin_amdgcn_global_atomic_fadd_f64(acc + static_cast<int64_t>(i) * q, s_row);
                    s_row += delta[i];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double yv = fma(-s_row, a.inv_taps, static_cast<double>(xo[i]));  // (:869 is the repair pass's)
                const TO yo = static_cast<TO>(yv);
                if constexpr (sizeof(TO) == 8) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, yo), rsrc_y, voff_y, soff_y + i * qy, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yo), rsrc_y, voff_y, soff_y + i * qy, 0);
                }
                s_row += delta[i];
            }
        }
        soff_y += static_cast<unsigned>(NGR) * qy;
        advance_slots();
    };

    TI pre_b[R];
    int mk = 0;
    while (mk < rows_total) {
        if (fast_ok && mk >= pair_lo && mk <= pair_hi) {
            // sample of (row, column 0 of the fill lanes) = s0 - B + row*q; of an output row = s0 + row*q
            soff_x = static_cast<unsigned>((s0 - B - a.buf_first + static_cast<int64_t>(mk + 2 * NGR - a_lo + g * R) * q) * kEl);
            soff_y = static_cast<unsigned>((s0 - a.out_first + static_cast<int64_t>(mk + g * R) * q) *
                                           static_cast<int64_t>(sizeof(TO)));
            do {
                fast_iteration(pre_a, pre_b);
                fast_iteration(pre_b, pre_a);
                mk += 2 * NGR;
            } while (mk <= pair_hi);
        } else {
            iteration(mk, pre_a, pre_b);
            if (mk + NGR < rows_total) iteration(mk + NGR, pre_b, pre_a);
            mk += 2 * NGR;
        }
    }
}

[[maybe_unused]] int pad_half(int64_t n) {
    for (int d : {4, 8, 12, 14, 16, 20, 24, 32})
        if (n <= d) return d;
    return 0;
}

struct Split {
    int a, b;
};
inline Split split(int64_t u, int64_t q) {  // nearest multiple: b in (-q/2, q/2]
    const int64_t a = static_cast<int64_t>(std::floor(static_cast<double>(u) / static_cast<double>(q) + 0.5));
    return {static_cast<int>(a), static_cast<int>(u - a * q)};
}
inline Split split_floor(int64_t u, int64_t q) {  // wrap form: b in [0, q)
    const int64_t a = u >= 0 ? u / q : -((-u + q - 1) / q);
    return {static_cast<int>(a), static_cast<int>(u - a * q)};
}

size_t lds_bytes(const PhaseGeom &g, size_t el) {
    return align16(static_cast<size_t>(g.q + 2 * g.guard) * g.rs * el) +
           static_cast<size_t>(2 * g.n_groups * g.qp) * sizeof(double);
}

}  // namespace

#ifdef PARRM_PHASE_WITH_PLAN
void plan_phase(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *plan, std::vector<int32_t> *table) {
    auto tap_at = [&](int64_t w) -> int { return (w >= -hw && w <= hw) ? tap[w + hw] : 0; };
    double best_cost = 1e300;
    int64_t best_q = 0;
    std::vector<int64_t> taps_at;  // offsets of the taps, ascending
    for (int64_t w = -hw; w <= hw; ++w)
        if (tap[w + hw]) taps_at.push_back(w);
    for (int64_t q = 64; q <= 512; ++q) {
        // positions that matter: the taps themselves (u = w) and where a tap arrives (u = w - q)
        int64_t nd = 0;
        int guard = 0;
        bool ok = true;
        for (size_t k = 0; k < taps_at.size() && ok; ++k) {
            const int64_t w = taps_at[k];
            nd += tap_at(w + q) == 0;                                    // u = w: tap leaves
            guard = std::max(guard, std::abs(split(w, q).b));
            if (tap_at(w - q) == 0) {                                    // u = w - q: tap arrives
                ++nd;
                guard = std::max(guard, std::abs(split(w - q, q).b));
            }
            ok = guard <= kMaxGuard;
        }
        if (!ok || nd == 0 || (nd & 1)) continue;
        const int dp = pad_half(nd / 2);
        if (dp == 0) continue;
        const int64_t qp = (q + 2 * guard + 63) / 64 * 64;
        // per output: 2*dp taps at (1 + R)/R vector instructions (R ~ 3) + ~20 for the epilogue
        const double cost = (2.0 * dp * 1.34 + 20.0 + 0.2 * guard) * static_cast<double>(qp) / static_cast<double>(q);
        if (cost < best_cost) {
            best_cost = cost;
            best_q = q;
        }
    }
    plan->phase = PhaseGeom{};
    if (getenv("PARRM_PHASE_FORCE_WRAP")) {  // tuning knob: wrap form even where the guarded one fits
        best_q = 0;
        best_cost = 1e300;
    }
    // Short periods: no stride keeps every tap within kMaxGuard residues of its multiples.  The wrap
    // form (b in [0, q), per-lane wrap to the previous row) takes any stride whose delta list is short.
    // Both searches run; the cheaper estimate wins (the guarded form where it exists with a short delta
    // list, e.g. the 22 kHz / 130 Hz comb; the wrap form when the guarded delta list is long).
    bool wrap = false;
    {
        const char *force = getenv("PARRM_PHASE_Q");  // tuning knob: the wrap form's stride
        const int64_t fq = force ? atoll(force) : 0;
        if (fq) best_cost = 1e300;
        for (int64_t q = 32; q <= 512; ++q) {  // (strides below one wave: integer periods, whose delta list grows with q)
            if (fq && q != fq) continue;
            int64_t both = 0;
            for (const int64_t w : taps_at) both += tap_at(w + q);
            const int64_t nd = 2 * (static_cast<int64_t>(taps_at.size()) - both);
            if (nd == 0) continue;
            const int dp = pad_half(nd / 2);
            if (dp == 0) continue;
            const int64_t qp = (q + 63) / 64 * 64;
            // two more vector instructions per tap than the guarded form (wrap bit, multiply-add)
            const double cost = (2.0 * dp * 2.0 + 20.0) * static_cast<double>(qp) / static_cast<double>(q);
            if (cost < best_cost) {
                best_cost = cost;
                best_q = q;
                wrap = true;
            }
        }
    }
    if (best_q == 0) return;
    const int64_t q = best_q;
    auto split = [&](int64_t u, int64_t qq) -> Split { return wrap ? split_floor(u, qq) : parrm_filter::split(u, qq); };
    PhaseGeom g{};
    g.q = static_cast<int32_t>(q);
    g.wrap = wrap ? 1 : 0;
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
    g.guard = wrap ? 0 : std::max(guard, 1);
    g.d_pad = pad_half(static_cast<int64_t>(plus.size()));
    g.n_runs = static_cast<int32_t>(runs.size() / 3);
    // the 2*guard halo columns are owned by the idle lanes of the last wave of each row group
    g.qp = static_cast<int32_t>((q + 2 * g.guard + 63) / 64 * 64);
    g.a_lo = std::min(a_lo, 0);
    g.a_hi = std::max(a_hi, 0) + (wrap ? 1 : 0);  // wrapped lanes reach one row further back
    // (groups, rows per thread), best first.  Measured on 256 x 10M f64, q = 169: (4,2) 10.2 ms,
    // (2,3) 10.3, (2,4) 10.4, (2,2) 11.2, (3,2) 11.5, (3,3) 11.9 -- 24 waves per CU (two 12-wave
    // workgroups, three waves per SIMD each) hide more latency than the extra instructions of R = 2 cost.
    int kShapes[8][2] = {{4, 2}, {2, 4}, {2, 3}, {2, 2}, {3, 2}, {1, 4}, {1, 3}, {1, 2}};
    if (wrap) {
        // wrap form: delta lists are short and strides often large; two-row shapes first (measured:
        // q = 500 as (2,4) 20 ms, as (1,2) 14 ms; q = 201 as (2,4) 20 ms, q = 197 as (2,3) 13.6 ms)
        const int order[8][2] = {{4, 2}, {2, 2}, {3, 2}, {1, 2}, {2, 3}, {1, 3}, {2, 4}, {1, 4}};
        for (int i = 0; i < 8; ++i) {
            kShapes[i][0] = order[i][0];
            kShapes[i][1] = order[i][1];
        }
    }
    if (const char *env = getenv("PARRM_PHASE_SHAPE")) {  // tuning knob: "NG,R" tried first
        int ng = 0, rr = 0;
        if (sscanf(env, "%d,%d", &ng, &rr) == 2 && ng >= 1 && ng <= 5 && rr >= 2 && rr <= 4) {
            kShapes[7][0] = kShapes[0][0];
            kShapes[7][1] = kShapes[0][1];
            kShapes[0][0] = ng;
            kShapes[0][1] = rr;
        }
    }
    bool placed = false;
    for (int pass = 0; pass < 2 && !placed; ++pass) {
        for (const auto &sh : kShapes) {
            g.n_groups = sh[0];
            g.rows = sh[1];
            const int ngr = g.n_groups * g.rows;
            // rows resident at once: the taps' reach around the NG*R rows being computed, plus the
            // NG*R rows published meanwhile
            g.m_slots = 2 * ngr + (g.a_hi - g.a_lo);
            g.rs = g.m_slots + g.rows - 1 + (wrap ? 1 : 0);  // + mirrored head, + the slot in front (wrap)
            if ((g.rs & 1) == 0) ++g.rs;
            if (g.n_groups * g.qp > 1024) continue;
            if (lds_bytes(g, sizeof(double)) <= (pass == 0 ? kLdsTwoBlocks : kLdsOneBlock)) {
                placed = true;
                break;
            }
        }
    }
    if (!placed) return;
    // tap table, one row per value of slot(m - a_lo): element offset of every delta tap relative to
    // this lane's (residue, slot 0) cell.  Padding: the same real tap appended to both signs cancels
    // (n_plus == n_minus always, because shifting the tap set preserves its size).
    table->clear();
    auto pick = [&](const std::vector<Split> &v, int t) { return t < (int)v.size() ? v[t] : plus[0]; };
    for (int sl = 0; sl < g.m_slots; ++sl) {
        for (int sign = 0; sign < 2; ++sign) {
            for (int t = 0; t < g.d_pad; ++t) {
                const Split s = pick(sign == 0 ? plus : minus, t);
                int slot = sl - (s.a - g.a_lo);
                if (slot < 0) slot += g.m_slots;
                table->push_back(slot - s.b * g.rs);
            }
        }
    }
    for (size_t k = 0; k < runs.size(); k += 3) {
        table->push_back(runs[k] - g.a_lo);
        table->push_back(runs[k + 1]);
        table->push_back(runs[k + 2]);
    }
    if (wrap) {  // residue offsets of the delta taps, for the per-lane wrap masks
        for (int sign = 0; sign < 2; ++sign)
            for (int t = 0; t < g.d_pad; ++t) table->push_back(pick(sign == 0 ? plus : minus, t).b);
    }
    // Packed float32 form (guarded, two rows per thread): its own shape -- two ring copies must fit next to
    // the exchange area -- and a table of BYTE offsets into the copies: copy E holds row slot s at element
    // s, copy O at element s + 1; a tap's pair of rows (s, s + 1) is read from the copy where it starts at
    // an even element.
    plan->phase_pack = PhaseGeom{};
    if (!wrap && !getenv("PARRM_NO_F32_PACK")) {
        PhaseGeom pk = g;
        pk.rows = 2;
        for (int ng : {4, 3, 2, 1}) {
            pk.n_groups = ng;
            pk.m_slots = 2 * ng * 2 + (g.a_hi - g.a_lo);
            // copy E uses elements 0 .. M (mirrored head at M); copy O holds row s at element s + 1 and is read
            // for odd s only: elements up to M + 1, or up to M when M is odd (the pair starting at the last
            // slot M - 1 is then an even one and comes from copy E)
            int rs2 = (pk.m_slots & 1) ? pk.m_slots + 1 : pk.m_slots + 2;
            while ((rs2 & 1) || ((rs2 / 2) & 1) == 0) ++rs2;  // even, half of it odd
            pk.rs = rs2;
            const size_t copy_bytes = align16(static_cast<size_t>(g.q + 2 * g.guard) * rs2 * sizeof(float));
            const size_t total = 2 * copy_bytes + static_cast<size_t>(2 * ng * g.qp) * sizeof(double);
            if (ng * g.qp > 1024 || total > kLdsTwoBlocks) continue;
            pk.copy_o_bytes = static_cast<int32_t>(copy_bytes);
            pk.tab_off = static_cast<int32_t>(table->size());
            for (int sl = 0; sl < pk.m_slots; ++sl) {
                for (int sign = 0; sign < 2; ++sign) {
                    for (int t = 0; t < g.d_pad; ++t) {
                        const Split s = pick(sign == 0 ? plus : minus, t);
                        int slot = sl - (s.a - g.a_lo);
                        if (slot < 0) slot += pk.m_slots;
                        const bool odd = slot & 1;
                        const int element = -s.b * rs2 + (odd ? slot + 1 : slot);
                        table->push_back(element * static_cast<int>(sizeof(float)) + (odd ? pk.copy_o_bytes : 0));
                    }
                }
            }
            for (size_t k = 0; k < runs.size(); k += 3) {
                table->push_back(runs[k] - g.a_lo);
                table->push_back(runs[k + 1]);
                table->push_back(runs[k + 2]);
            }
            plan->phase_pack = pk;
            break;
        }
    }
    plan->phase = g;
}
#endif  // PARRM_PHASE_WITH_PLAN

template <typename TI, typename TO, int DP, int MODE>
static void (*pick_phase_kernel(bool packed, int rows, bool wrap))(FilterArgs, PhaseGeom) {
    switch (packed ? 0 : rows * 2 + (wrap ? 1 : 0)) {
        case 0:
            if constexpr (sizeof(TI) == 4) return filter_phase_kernel<TI, TO, DP, 2, false, sizeof(TO) == 4 ? 2 : 1, MODE>;
            break;
        case 4: return filter_phase_kernel<TI, TO, DP, 2, false, 0, MODE>;
        case 5: return filter_phase_kernel<TI, TO, DP, 2, true, 0, MODE>;
        case 6: return filter_phase_kernel<TI, TO, DP, 3, false, 0, MODE>;
        case 7: return filter_phase_kernel<TI, TO, DP, 3, true, 0, MODE>;
        case 8: return filter_phase_kernel<TI, TO, DP, 4, false, 0, MODE>;
        case 9: return filter_phase_kernel<TI, TO, DP, 4, true, 0, MODE>;
        default: break;
    }
    return nullptr;
}

template <typename TI, typename TO, int DP>
static int launch_phase_r(const parrm_filter_plan *plan, FilterArgs *args, PhaseGeom g, hipStream_t stream, int mode) {
    FilterArgs &a = *args;
    void (*kern)(FilterArgs, PhaseGeom) = nullptr;
    const bool packed = g.copy_o_bytes > 0;  // float32 recordings: the packed geometry (launch_phase chose it)
    if (mode == 1) {
        if constexpr (sizeof(TO) == 8) kern = pick_phase_kernel<TI, TO, DP, 1>(packed, g.rows, g.wrap != 0);
    } else {
        kern = pick_phase_kernel<TI, TO, DP, 0>(packed, g.rows, g.wrap != 0);
    }
    if (!kern) {
        parrm::set_error("filter: unsupported phase kernel shape (rows %d, mode %d)", g.rows, mode);
        return PARRM_ERR_INVALID;
    }
    const int ngr = g.n_groups * g.rows;
    // Stretch length.  A stretch costs its samples plus a fixed prologue (the ring fill of 2*hw + NG*R*q
    // samples and one full tap evaluation per thread: ~16 K samples' worth, fitted on 64 x 1 M).  512
    // workgroups are resident at once (two per CU), so a launch runs in ceil(blocks / 512) rounds of one
    // stretch each.  Long recordings: ~256 K-sample stretches (128 K measured 1-1.5 % slower, 512 K 2 %
    // slower: with many rounds the tail evens out, and longer stretches only lose balance).  Short ones
    // (e.g. 64 ch x 1 M): pick the stretch count per channel that minimises rounds x (stretch + prologue)
    // -- round 1 always cut down to >= 2048 workgroups, i.e. 31 K-sample stretches at 64 x 1 M, where the
    // prologue was a third of the work.
    int64_t target = 262144;
    if (const char *env = getenv("PARRM_STRETCH_SAMPLES")) target = std::max<int64_t>(atoll(env), g.q);  // tuning knob
    int64_t rows = std::max<int64_t>(ngr, (target / g.q) / ngr * ngr);
    auto blocks_for = [&](int64_t r) { return a.plan_chans * ((a.out_len + r * g.q - 1) / (r * g.q)); };
    if (blocks_for(rows) < 4096 && !getenv("PARRM_STRETCH_SAMPLES")) {
        const double prologue = 16384.0;
        const int64_t resident = 2 * static_cast<int64_t>(parrm::device_cu_count());  // two workgroups per CU
        double best = 1e300;
        int64_t best_rows = rows;
        const int64_t max_stretches = std::max<int64_t>(1, a.out_len / (static_cast<int64_t>(4) * ngr * g.q));
        for (int64_t per_chan = 1; per_chan <= std::min<int64_t>(max_stretches, 4096); ++per_chan) {
            const int64_t r = ((a.out_len + per_chan - 1) / per_chan + g.q - 1) / g.q;      // rows per stretch
            const int64_t rr = (r + ngr - 1) / ngr * ngr;                                     // whole iterations
            const int64_t blocks = blocks_for(rr);
            const int64_t rounds = (blocks + resident - 1) / resident;
            // a last, partly filled round costs a whole round; beyond a few rounds the dispatcher evens things out
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
    const size_t lds = packed ? 2 * align16(static_cast<size_t>(g.q + 2 * g.guard) * g.rs * sizeof(TI)) +
                                    static_cast<size_t>(2 * g.n_groups * g.qp) * sizeof(double)
                              : lds_bytes(g, sizeof(TI));
    PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(g.n_groups * g.qp), lds, stream, a, g);
    PARRM_HIP_CHECK(hipGetLastError());
    (void)plan;
    return PARRM_OK;
}

template <typename TI, typename TO>
int launch_phase(const parrm_filter_plan *plan, FilterArgs *a, hipStream_t stream, int mode) {
    PhaseGeom g = plan->phase;
    g.tab = plan->d_phase_tab;
    g.copy_o_bytes = 0;
    // float32 recordings: the packed form where the plan has one.  float32 output sums the taps of a row in
    // float32 (PACK = 2: 8.6 -> 6.0 ms on 256 ch x 10 M); float64 output keeps float64 sums (PACK = 1: same
    // values, half the LDS reads, 9.2 -> 8.9 ms).  PARRM_NO_F32_PACK at plan time turns both off.
    if constexpr (sizeof(PARRM_PHASE_TI) == 4) {
        if (plan->phase_pack.n_groups > 0) {
            g = plan->phase_pack;
            g.tab = plan->d_phase_tab + g.tab_off;
        }
    }
    if (const char *env = getenv("PARRM_DEBUG_FLAGS")) g.debug = atoi(env) & 31;  // profiling ablations only
    if (mode == 2) {  // MODE 1 kernels, storing instead of adding
        g.debug |= 32;
        mode = 1;
    }
    switch (g.d_pad) {
        case 4: return launch_phase_r<TI, TO, 4>(plan, a, g, stream, mode);
        case 8: return launch_phase_r<TI, TO, 8>(plan, a, g, stream, mode);
        case 12: return launch_phase_r<TI, TO, 12>(plan, a, g, stream, mode);
        case 14: return launch_phase_r<TI, TO, 14>(plan, a, g, stream, mode);
        case 20: return launch_phase_r<TI, TO, 20>(plan, a, g, stream, mode);
        case 16: return launch_phase_r<TI, TO, 16>(plan, a, g, stream, mode);
        case 24: return launch_phase_r<TI, TO, 24>(plan, a, g, stream, mode);
        case 32: return launch_phase_r<TI, TO, 32>(plan, a, g, stream, mode);
        default: parrm::set_error("filter: no phase kernel for %d delta taps per sign", g.d_pad); return PARRM_ERR_INVALID;
    }
}

template int launch_phase<PARRM_PHASE_TI, PARRM_PHASE_TO>(const parrm_filter_plan *, FilterArgs *, hipStream_t, int);

}  // namespace parrm_filter
