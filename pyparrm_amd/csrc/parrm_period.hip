// find_period on gfx950: the three device stages behind parrm.py:272-280 and :552-632.
//
//   absdiff_*        one streaming pass: scale[c] = mean_i |x[c,i+1]-x[c,i]|        (:274-275)
//   gather_kernel    Y[j][c] = clip((x[c,idx_j+1]-x[c,idx_j])/scale[c], +-ob)        (:274-278,:590)
//   fit_accum_kernel per (period, sample slice): the Gram blocks  W'[Y | W]  of the harmonic
//                    regression, W = [1, sin(k a), cos(k a)], a = (idx+1)*(2pi/T)    (:619-626)
//   fit_reduce_kernel / fit_solve_kernel: sum the slices, LU-solve the normal equations, turn
//                    (G, R, y'y) into the channel-averaged regularised error         (:585-597,:626-632)
//
// The regression is a skinny f64 contraction (K <= 47 rows, n <= 25001 samples, C+K columns), bound
// by FP64 arithmetic, not by HBM: Y (<= 51 MB) stays in L2 / Infinity Cache.  It runs on the matrix
// cores (fit_accum_mfma_kernel, v_mfma_f64_16x16x4_f64: the FP64 matrix and vector peaks are equal on
// MI355X, but the MFMA form needs far fewer operand fetches and leaves the vector ALU free);
// fit_accum_kernel is the vector-ALU form for operand layouts the MFMA kernel cannot take.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "parrm_common.h"
#include "parrm_nm_core.h"

namespace {

// ------------------------------------------------------------------------------------ a3
constexpr int kStatChunk = 32768;  // diffs per workgroup

template <typename T>
__global__ void __launch_bounds__(256) absdiff_partial_kernel(const T *x, int64_t n_samples, int64_t ldx,
                                                               int64_t n_chunk, double *partial) {
    const int64_t c = blockIdx.x / n_chunk;
    const int64_t k = blockIdx.x - c * n_chunk;
    const T *row = x + c * ldx;
    const int64_t lo = k * kStatChunk;
    int64_t hi = lo + kStatChunk;
    if (hi > n_samples - 1) hi = n_samples - 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {  // 4 independent loads in flight per lane
        const T d0 = row[i + 1] - row[i];
        const T d1 = row[i + 257] - row[i + 256];
        const T d2 = row[i + 513] - row[i + 512];
        const T d3 = row[i + 769] - row[i + 768];
        a0 += fabs(static_cast<double>(d0));
        a1 += fabs(static_cast<double>(d1));
        a2 += fabs(static_cast<double>(d2));
        a3 += fabs(static_cast<double>(d3));
    }
    for (; i < hi; i += 256) a0 += fabs(static_cast<double>(static_cast<T>(row[i + 1] - row[i])));
    double v = parrm::wave_sum((a0 + a1) + (a2 + a3));
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void __launch_bounds__(64) absdiff_final_kernel(const double *partial, int64_t n_chunk,
                                                            int64_t n_samples, double *scale) {
    const int64_t c = blockIdx.x;
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < n_chunk; k += 64) acc += partial[c * n_chunk + k];
    acc = parrm::wave_sum(acc);
    if (threadIdx.x == 0) scale[c] = acc / static_cast<double>(n_samples - 1);
}

// ------------------------------------------------------------------------------------ a3 o a4
template <typename T>
__global__ void __launch_bounds__(256) gather_kernel(const T *x, int64_t n_chans, int64_t ldx,
                                                      const int64_t *idx, int64_t n_idx, const double *scale,
                                                      double ob, double *y, int64_t ldy) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= n_idx) return;
    const int64_t i = idx[j];
    for (int64_t c = blockIdx.y; c < n_chans; c += gridDim.y) {
        const T *row = x + c * ldx;
        const T d = row[i + 1] - row[i];  // np.diff keeps the input dtype (:274)
        const double v = static_cast<double>(d) / scale[c];             // :275
        y[j * ldy + c] = v != v ? v : fmin(fmax(v, -ob), ob);           // :276-278 (np.clip keeps NaN)
    }
}

// ------------------------------------------------------------------------------------ a6 / a7
constexpr int kTJ = 16;      // samples per LDS tile
constexpr int kYCols = 256;  // data columns per workgroup
constexpr int kNCol = 320;   // + up to 64 columns holding W itself (-> Gram matrix)

// Block coordinates of a kernel body.  The fit kernels below are written as __device__ bodies over explicit block
// coordinates with two kinds of __global__ wrappers: the plain one (coordinates = blockIdx) and the "gang" one of
// parrm_fit_errors_multi, where blockIdx.z picks one of several independent problems from a device table.
struct Blk {
    int x, y, z, nz;
};
__device__ inline Blk this_block() {
    return Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), static_cast<int>(blockIdx.z), static_cast<int>(gridDim.z)};
}
constexpr int kMaxBw = 23;

// An optimiser batch's candidate periods travel as a kernel ARGUMENT of the design-matrix kernel (<= kSmallPeriods of
// them): the H2D copy they used to take was a blit kernel of its own in front of every batch (~4 us + ~4.5 us of
// dispatch gap + the host's hipMemcpyAsync, 76 times per search).
// The struct is the kernels' FIRST parameter and is never named in their bodies: indexing a by-value aggregate with a
// run-time index makes hipcc copy it into scratch in every thread (measured: 520 bytes of private segment per lane,
// find_period 25 -> 33 ms); the values are read where the dispatch packet put them, at the start of the kernarg segment.
constexpr int kSmallPeriods = 64;
struct SmallPeriods {
    double p[kSmallPeriods];
};
__device__ __forceinline__ const double *small_periods_in_kernarg() {
    return (const double *)__builtin_amdgcn_kernarg_segment_ptr();  // (a C cast: it crosses the address space)
}

__host__ __device__ inline int kp_for(int bw) {  // padded row count: 4 waves x KT rows
    const int K = 2 * bw + 1;
    return K <= 12 ? 12 : (K <= 24 ? 24 : 48);
}

// ---- design matrix ---------------------------------------------------------------------------
// W[p][k][j] (k-major, j padded to a multiple of kTJ with zeros): 1, sin(a), cos(a), sin(2a), ...
// with a = (idx_j + 1) * (2*pi / T_p) rounded as the reference rounds it (parrm.py:619).
//   exact == 0: one sincos(a) per sample, higher harmonics by the angle-addition recurrence
//               (error ~ k*eps; the reference's own sin(fl(k*a)) carries ~eps*k*a/2 ~ 1e-10 of
//               argument rounding at a ~ 1e5 rad, so the two agree to that level);
//   exact == 1: sincos(fl(k*a)) per harmonic, the reference's operation order (:622-623).
__global__ void __launch_bounds__(256) fit_trig_kernel(const SmallPeriods, const int64_t *idx, int n_idx, int n_pad,
                                                        const double *periods, int bw, int kpc, int exact,
                                                        double *W, int by_value) {
    if (by_value) periods = small_periods_in_kernarg();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (j >= n_pad) return;
    double *w = W + (static_cast<int64_t>(p) * kpc) * n_pad + j;
    const int K = 2 * bw + 1;
    if (j >= n_idx) {
        for (int k = 0; k < kpc; ++k) w[static_cast<int64_t>(k) * n_pad] = 0.0;
        return;
    }
    const double w0 = 6.283185307179586 / periods[p];             // (2 * np.pi / period)
    const double ang = static_cast<double>(idx[j] + 1) * w0;      // (indices + 1) * ...
    w[0] = 1.0;
    double s1, c1;
    sincos(ang, &s1, &c1);
    double sk = s1, ck = c1;
    for (int k = 1; k <= bw; ++k) {
        if (exact && k > 1) sincos(static_cast<double>(k) * ang, &sk, &ck);
        w[static_cast<int64_t>(2 * k - 1) * n_pad] = sk;
        w[static_cast<int64_t>(2 * k) * n_pad] = ck;
        const double sn = fma(sk, c1, ck * s1);
        const double cn = fma(ck, c1, -(sk * s1));
        sk = sn;
        ck = cn;
    }
    for (int k = K; k < kpc; ++k) w[static_cast<int64_t>(k) * n_pad] = 0.0;
}

// The same values in the sample-major "stacked" layout of fit_accum_mfma_kernel:
// Ws[p / NCB][j][48], stacked row (p % NCB)*kpc + k (a 16-row x 4-sample MFMA operand is then four
// 128-byte segments).  One workgroup = 64 samples x the NCB candidates of one stack (wave q =
// candidate q): the 64 x 48 block is assembled in LDS and written out as one contiguous 24 KB run.
// Candidates >= n_periods and samples >= n_idx are zeros.
__device__ __forceinline__ void fit_trig_stacked_body(const int64_t *idx, int n_idx, int n_pad, const double *periods,
                                                      int n_periods, int bw, int kpc, int exact, double *Ws, const Blk blk) {
    constexpr int KS = 48, TS = 49;  // padded tile row: conflict-free b64 writes at one row per lane
    __shared__ double tile[64 * TS];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int ncb = KS / kpc;
    const int j = blk.x * 64 + lane;
    const int p = blk.y * ncb + q;
    const int K = 2 * bw + 1;
    double *t = tile + lane * TS + q * kpc;
    if (j >= n_idx || p >= n_periods) {
        for (int k = 0; k < kpc; ++k) t[k] = 0.0;
    } else {
        const double w0 = 6.283185307179586 / periods[p];         // (2 * np.pi / period)
        const double ang = static_cast<double>(idx[j] + 1) * w0;  // (indices + 1) * ...
        t[0] = 1.0;
        double s1, c1;
        sincos(ang, &s1, &c1);
        double sk = s1, ck = c1;
        for (int k = 1; k <= bw; ++k) {
            if (exact && k > 1) sincos(static_cast<double>(k) * ang, &sk, &ck);
            t[2 * k - 1] = sk;
            t[2 * k] = ck;
            const double sn = fma(sk, c1, ck * s1);
            const double cn = fma(ck, c1, -(sk * s1));
            sk = sn;
            ck = cn;
        }
        for (int k = K; k < kpc; ++k) t[k] = 0.0;
    }
    __syncthreads();
    const int rows = min(64, n_pad - blk.x * 64);  // n_pad is a multiple of 16
    double *dst = Ws + (static_cast<int64_t>(blk.y) * n_pad + static_cast<int64_t>(blk.x) * 64) * KS;
    for (int e = threadIdx.x; e < rows * KS; e += blockDim.x) dst[e] = tile[(e / KS) * TS + (e % KS)];
}
__global__ void __launch_bounds__(256) fit_trig_stacked_kernel(const SmallPeriods, const int64_t *idx, int n_idx, int n_pad,
                                                                const double *periods, int n_periods, int bw,
                                                                int kpc, int exact, double *Ws, int by_value) {
    fit_trig_stacked_body(idx, n_idx, n_pad, by_value ? small_periods_in_kernarg() : periods, n_periods, bw, kpc, exact, Ws,
                          this_block());
}

// The PACKED stacked layout (GM = 2 of fit_accum_mfma_body): global row K p + k for row k of candidate p, stack
// s = global rows [48 s, 48 s + 48).  One workgroup = 64 samples x one stack; wave q = the q-th candidate with rows in
// the stack (<= 6 / 4 / 3 of them at K = 11 / 21 / 41: a candidate that straddles two stacks is evaluated by both
// workgroups -- the recurrence is cheap beside the 24 KB the workgroup writes, and the block leaves as ONE contiguous
// run as in the padded layout; writing per candidate group instead, <= 2 runs of 41-44 rows per sample, measured
// twice the time: partial lines shared with the neighbouring group).  Candidates >= n_periods and samples >= n_idx
// are zeros.
constexpr int packed_trig_waves(int K) { return (48 + K - 2) / K + 1; }  // most candidates a 48-row window can touch
__device__ __forceinline__ void fit_trig_packed_body(const int64_t *idx, int n_idx, int n_pad, const double *periods,
                                                     int n_periods, int bw, int exact, double *Ws, const Blk blk) {
    constexpr int KS = 48, TS = 49;
    __shared__ double tile[64 * TS];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int K = 2 * bw + 1;
    const int g0 = KS * blk.y;       // first global row of the stack
    const int p = g0 / K + q;        // this wave's candidate
    const int k_lo = max(g0 - p * K, 0), k_hi = min(g0 + KS - p * K, K);  // its rows [k_lo, k_hi) lie in the stack
    const int j = blk.x * 64 + lane;
    if (k_lo < k_hi) {
        double *t = tile + lane * TS + (p * K - g0);  // t[k] = row k of the candidate (k_lo <= k < k_hi)
        if (j >= n_idx || p >= n_periods) {
            for (int k = k_lo; k < k_hi; ++k) t[k] = 0.0;
        } else {
            const double w0 = 6.283185307179586 / periods[p];         // (2 * np.pi / period)
            const double ang = static_cast<double>(idx[j] + 1) * w0;  // (indices + 1) * ...
            if (k_lo == 0) t[0] = 1.0;
            double s1, c1;
            sincos(ang, &s1, &c1);
            double sk = s1, ck = c1;
            for (int k = 1; 2 * k - 1 < k_hi; ++k) {
                if (exact && k > 1) sincos(static_cast<double>(k) * ang, &sk, &ck);
                if (2 * k - 1 >= k_lo) t[2 * k - 1] = sk;
                if (2 * k >= k_lo && 2 * k < k_hi) t[2 * k] = ck;
                const double sn = fma(sk, c1, ck * s1);
                const double cn = fma(ck, c1, -(sk * s1));
                sk = sn;
                ck = cn;
            }
        }
    }
    __syncthreads();
    const int rows = min(64, n_pad - blk.x * 64);  // n_pad is a multiple of 16
    double *dst = Ws + (static_cast<int64_t>(blk.y) * n_pad + static_cast<int64_t>(blk.x) * 64) * KS;
    for (int e = threadIdx.x; e < rows * KS; e += blockDim.x) dst[e] = tile[(e / KS) * TS + (e % KS)];
}
__global__ void __launch_bounds__(384) fit_trig_packed_kernel(const SmallPeriods, const int64_t *idx, int n_idx, int n_pad,
                                                               const double *periods, int n_periods, int bw, int exact,
                                                               double *Ws, int by_value) {
    fit_trig_packed_body(idx, n_idx, n_pad, by_value ? small_periods_in_kernarg() : periods, n_periods, bw, exact, Ws,
                         this_block());
}

// ---- Gram blocks -----------------------------------------------------------------------------
// One workgroup = 4 waves x 12 stacked design rows = NCB = 48/KPC candidates (KPC = 12/24/48 padded
// rows per candidate), all data columns of one 256-channel block plus the 48 stacked W columns
// (-> each candidate's W'W falls out of the same pass), one slice of the samples.
// part[(((p*nz + zb)*nsplit + s)*(KPC+1) + row)*kNCol + col]; row KPC holds y'y per column; the
// Gram matrix of candidate p sits in columns kYCols + (p % NCB)*KPC + [0, K).
// Tiles of 16 samples go global -> registers (next tile, while the current one is being used) ->
// LDS; the inner loop is 60 FMAs per 17 LDS reads per lane.  (Feeding the 12 wave-uniform design
// values through scalar loads + scalar FMA operands instead of LDS was tried: 1.4x slower, the
// scalar loads could not be prefetched far enough within the SGPR budget.)
__global__ void __launch_bounds__(256) fit_accum_kernel(const double *Y, int64_t ldy, const double *W, int n_pad,
                                                         int n_idx, int n_chans, int n_periods, int kpc,
                                                         int nsplit, double *part) {
    constexpr int KT = 12, KS = 48;
    __shared__ double Wt[kTJ][KS];
    __shared__ __attribute__((aligned(16))) double Yt[kTJ][kNCol];
    const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 6;
    const int s = blockIdx.x, pb = blockIdx.y, zb = blockIdx.z, nz = gridDim.z;
    const int ncb = KS / kpc;

    double acc[KT][5];
    double yy[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        yy[i] = 0.0;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) acc[kk][i] = 0.0;
    }
    for (int it = tid; it < kTJ * 16; it += 256) Yt[it / 16][kYCols + KS + (it % 16)] = 0.0;  // cols 304..319

    const int tiles = n_pad / kTJ;
    const int per = (tiles + nsplit - 1) / nsplit;
    const int t_lo = s * per;
    const int t_hi = min(tiles, t_lo + per);

    // this thread's share of a tile: 16 Y elements (row j = it/256, column it%256) and 3 W elements
    // (stacked row r = it/16, sample j = it%16: contiguous in j in the k-major W)
    double ry[16], rw[3];
    auto fetch = [&](int t) {
        const int j0 = t * kTJ;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jj = j0 + i, gc = zb * kYCols + tid;
            ry[i] = (jj < n_idx && gc < n_chans) ? Y[static_cast<int64_t>(jj) * ldy + gc] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int it = tid + 256 * i, r = it >> 4, j = it & 15;
            const int pc = pb * ncb + r / kpc;
            rw[i] = pc < n_periods ? W[(static_cast<int64_t>(pc) * kpc + (r % kpc)) * n_pad + j0 + j] : 0.0;
        }
    };
    if (t_lo < t_hi) fetch(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
        __syncthreads();  // the previous tile is no longer being read
#pragma unroll
        for (int i = 0; i < 16; ++i) Yt[i][tid] = ry[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int it = tid + 256 * i, r = it >> 4, j = it & 15;
            Wt[j][r] = rw[i];
            Yt[j][kYCols + r] = rw[i];
        }
        __syncthreads();
        if (t + 1 < t_hi) fetch(t + 1);  // lands during the FMAs below
#pragma unroll 4
        for (int j = 0; j < kTJ; ++j) {
            double w[KT], y[5];
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) w[kk] = Wt[j][kg * KT + kk];
            {   // lane l: data columns {2l, 2l+1, 128+2l, 129+2l} as two full-rate ds_read_b128
                // (contiguous across the wave), stacked-W column 256+l as one ds_read_b64
                const double2 ya = *reinterpret_cast<const double2 *>(&Yt[j][2 * lane]);
                const double2 yb = *reinterpret_cast<const double2 *>(&Yt[j][128 + 2 * lane]);
                y[0] = ya.x;
                y[1] = ya.y;
                y[2] = yb.x;
                y[3] = yb.y;
                y[4] = Yt[j][kYCols + lane];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                yy[i] = fma(y[i], y[i], yy[i]);
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) acc[kk][i] = fma(w[kk], y[i], acc[kk][i]);
            }
        }
    }
    const int r0 = kg * KT;                 // first stacked row of this wave
    const int pc = pb * ncb + r0 / kpc;     // its candidate
    if (pc >= n_periods) return;
    const int k0 = r0 % kpc;
    double *out = part + ((static_cast<int64_t>(pc) * nz + zb) * nsplit + s) * (kpc + 1) * kNCol;
    const int col[5] = {2 * lane, 2 * lane + 1, 128 + 2 * lane, 129 + 2 * lane, kYCols + lane};
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
#pragma unroll
        for (int i = 0; i < 5; ++i) out[(k0 + kk) * kNCol + col[i]] = acc[kk][i];
    if (k0 == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) out[kpc * kNCol + col[i]] = yy[i];
    }
}

// ---- Gram blocks on the matrix cores ----------------------------------------------------------
// Same blocking and the same output as fit_accum_kernel, computed with v_mfma_f64_16x16x4_f64:
//   D[16 stacked rows][16 columns] += A[16 rows][4 samples] * B[4 samples][16 columns].
// Lane l = 16*s + c holds A[row c][sample s] and B[sample s][column c] (one f64 each), so
//   * a B operand is Y[j0+s][columns]: lane (s, c) takes the FOUR adjacent columns 64*wave + 4c + t
//     (two 16-byte loads; tile t of the wave then holds columns 4c + t, a permutation undone when
//     the block is written);
//   * an A operand is Ws[j0+s][16*rt + c]: four 128-byte segments of the sample-major stack;
//   * the Gram tiles need no extra loads: the B operand of W'W is the A operand of the other tile.
// Operands go global/L2 -> registers directly (no LDS, no barriers), through buffer descriptors
// whose range check supplies the zeros of the sample tail and of missing channel columns, four
// steps ahead of their use.  Per 4-sample step a wave issues 12 (+3 Gram) MFMAs for 7 loads.
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

// NW = waves per workgroup: 4 (256 data columns; the Gram row-tiles spread over waves 0-2), or 1 for
// recordings of <= 64 channels (one wave takes the 64 data columns and all nine Gram tiles: 21 MFMAs
// per step instead of 60 -- few-channel recordings are the common case in practice).
//
// GM = 2 (round 3): PACKED rows and the "special rows" form of W'W.
//  * Packed: the design rows of the candidates follow each other without padding -- global row K p + k for row k of
//    candidate p, stack s = global rows [48 s, 48 s + 48) -- so a candidate may straddle two stacks and no MFMA row is
//    spent on padding (GM = 1 pads a candidate to kpc = 12 / 24 / 48 rows: 8 / 12.5 / 15 % of the products at
//    K = 11 / 21 / 41).  An MFMA output row depends on its own A row only, so where a row sits changes no bit of it.
//  * Special rows: with W = [1, sin(k a), cos(k a)] every entry of W'W is a sum of two of the moments
//    C_m = sum_j cos(m a_j), S_m = sum_j sin(m a_j), m = 0 ... 2 bw (product-to-sum), and all of those follow from
//    THREE rows of the matrix: row 0 (m <= bw) and the rows of sin(bw a), cos(bw a)
//    (sin((bw+i) a) = sin(bw a) cos(i a) + cos(bw a) sin(i a), ...).  So per 16-column tile of the stack one extra
//    A operand -- the three rows of each of the <= 3 candidates that own columns of the tile, fetched with per-lane
//    row offsets (from the neighbouring stack where a candidate straddles) -- is multiplied with the tile: 3 MFMAs
//    per workgroup and step instead of 9 (per wave 13 instead of 15 at 256 channels, 6 instead of 12 at <= 16), and
//    the solvers rebuild the matrix (gram_from_special_rows).
// `kreal` = K = 2 bw + 1 (GM = 2 only; kpc is then only the row count of a candidate's result block).
template <int NW, int CT, int GM = 1>
__device__ __forceinline__ void fit_accum_mfma_body(const double *Y, int64_t ldy, const double *Ws, int n_pad, int n_idx,
                                                    int n_chans, int n_periods, int kpc, int nsplit, double *part,
                                                    const Blk blk, int kreal = 0) {
    static_assert(NW == 1 || NW == 2 || NW == 4, "one wave (<= 64 channels), two (<= 128) or four (256 per block)");
    static_assert(CT == 4 || ((CT == 1 || CT == 2) && NW == 1), "CT = 1, 2: one-wave forms for <= 16 / <= 32 channels");
    // Gram row-tiles per wave: NW = 4: waves 0-2 one each (wave 3 repeats tile 2, not stored); NW = 2: wave 0 tiles
    // 0 and 1, wave 1 tile 2 twice (the second not stored: straight-line code for both waves); NW = 1: all three
    constexpr int KS = 48, D = 4, NG = NW == 4 ? 1 : (NW == 2 ? 2 : 3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, sj = lane >> 4;
    const int sl = blk.x, pb = blk.y, zb = blk.z, nz = blk.nz;
    const int ncb = KS / kpc;

    const int tiles = n_pad / kTJ;
    const int per = (tiles + nsplit - 1) / nsplit;
    const int u_lo = sl * per * 4;                     // steps of 4 samples
    const int u_hi = min(tiles, sl * per + per) * 4;

    const int col0 = CT == 4 ? 64 * wv + 4 * c : CT * c;  // first of this lane's CT adjacent data columns
    // a quad may straddle n_chans: its extra columns come from the row's padding (or, with a tight
    // ldy, from the next row / the zero tail) and land in result columns that nothing reads
    const bool col_ok = zb * kYCols + col0 < n_chans;
    const unsigned step_y = static_cast<unsigned>(4 * ldy * 8), step_w = 4 * KS * 8;
    unsigned vy = col_ok ? static_cast<unsigned>(((static_cast<int64_t>(u_lo) * 4 + sj) * ldy + col0) * 8) : 0x80000000u;
    unsigned vw = static_cast<unsigned>(((u_lo * 4 + sj) * KS + c) * 8);
    // GM = 2: per column tile of the wave, lane c of a sample group fetches special row c % 3 (rows 0, K-2, K-1) of the
    // (c / 3)-th candidate that owns columns of the tile; the descriptor spans the stacks s-1 ... s+1
    constexpr int NS = GM == 2 ? NG : 0;  // column tiles of the special-row product per wave (as Gram row tiles before)
    const int K = kreal, g0 = KS * pb;    // (GM = 2) first global row of this stack
    const int n_stacks = GM == 2 ? (n_periods * K + KS - 1) / KS : 0;
    const int s_base = max(pb - 1, 0);
    auto tile_of = [&](int k) -> int {  // column tile k of this wave
        return NW == 4 ? min(wv, 2) : (NW == 2 ? (wv == 0 ? k : 2) : k);
    };
    unsigned vs[NS == 0 ? 1 : NS];
    if constexpr (GM == 2) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int gl = g0 + 16 * tile_of(k);  // first global row (= W'W column) of the tile
            const int ci = c / 3, t = c - 3 * ci;
            const int pc = gl / K + ci;
            const bool own = pc <= (gl + 15) / K && pc < n_periods;
            const int gsp = own ? pc * K + (t == 0 ? 0 : K - 3 + t) : gl;
            const int s2 = gsp / KS, r2 = gsp - KS * s2;
            vs[k] = static_cast<unsigned>(((static_cast<int64_t>(s2 - s_base) * n_pad + u_lo * 4 + sj) * KS + r2) * 8);
        }
    }

    d4_t acc[3][CT], gacc[GM == 2 ? 1 : NG][3], gsp[NS == 0 ? 1 : NS];
    double yy[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) yy[t] = 0.0;
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int gr = 0; gr < NG; ++gr)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) gacc[GM == 2 ? 0 : gr][ct] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < (NS == 0 ? 1 : NS); ++k) gsp[k] = d4_t{0.0, 0.0, 0.0, 0.0};
    // Operand requests as inline assembly with hand-counted waits.  With compiler-visible loads hipcc either gathers
    // the four steps' requests into one batch per iteration and rotates the operands through copies, or -- requests
    // pinned behind each step's products -- cannot count across the loop's back edge and drains the queue
    // (vmcnt(0)) at the top of every iteration: either way the prefetch depth is given up once per iteration and the
    // matrix pipe idles for a memory latency (the launches sat at 87 % of the MFMA bound).  Loads retire in issue
    // order among themselves and the loop issues nothing else, so "at most 3 steps' requests outstanding" means the
    // oldest step's operands have landed.
    constexpr int NYL = CT == 4 ? 2 : 1;                    // data requests per step
    constexpr int LPS = NYL + 3 + (GM == 2 ? NS : 0);       // requests per step
    // (raw buffer descriptors: base, 48-bit address | stride 0, byte count, the flags of make_buffer_rsrc above)
    auto desc_of = [](const double *base, int bytes) -> u4_t {
        const unsigned long long p = reinterpret_cast<unsigned long long>(base);
        return u4_t{static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(p))),
                    static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(p >> 32))) & 0xffffu,
                    static_cast<unsigned>(__builtin_amdgcn_readfirstlane(bytes)), 0x00020000u};
    };
    const u4_t dy = desc_of(Y + static_cast<int64_t>(zb) * kYCols,
                            static_cast<int>((static_cast<int64_t>(n_idx) * ldy - static_cast<int64_t>(zb) * kYCols) * 8));
    const u4_t dw = desc_of(Ws + static_cast<int64_t>(pb) * n_pad * KS, n_pad * KS * 8);
    const u4_t dsp = desc_of(Ws + static_cast<int64_t>(s_base) * n_pad * KS,
                             GM == 2 ? min(3, n_stacks - s_base) * n_pad * KS * 8 : 0);
    u4_t qy[D][NYL];                          // data: CT = 4 two quads of 16 bytes, CT = 2 one; CT = 1: .x/.y only
    u2_t qa[D][3], qs[D][NS == 0 ? 1 : NS];   // design rows of the three row tiles; special rows
    auto fetch = [&](int d) {
        if constexpr (CT == 4) {
            asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16"
                         : "=&v"(qy[d][0]), "=&v"(qy[d][1]) : "v"(vy), "s"(dy) : "memory");
        } else if constexpr (CT == 2) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(qy[d][0]) : "v"(vy), "s"(dy) : "memory");
        } else {
            u2_t one;
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(one) : "v"(vy), "s"(dy) : "memory");
            qy[d][0].x = one.x;
            qy[d][0].y = one.y;
        }
        asm volatile("buffer_load_dwordx2 %0, %3, %4, 0 offen\n\tbuffer_load_dwordx2 %1, %3, %4, 0 offen offset:128\n\t"
                     "buffer_load_dwordx2 %2, %3, %4, 0 offen offset:256"
                     : "=&v"(qa[d][0]), "=&v"(qa[d][1]), "=&v"(qa[d][2]) : "v"(vw), "s"(dw) : "memory");
        if constexpr (GM == 2) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(qs[d][k]) : "v"(vs[k]), "s"(dsp) : "memory");
                vs[k] += step_w;
            }
        }
        if (col_ok) vy += step_y;
        vw += step_w;
    };
    // the step's operands have landed once at most the three younger steps' requests are outstanding; the "+v" ties
    // make every use of them depend on this wait
    auto landed = [&](int d) {
        if constexpr (CT == 4)
            asm volatile("s_waitcnt vmcnt(%5)" : "+v"(qy[d][0]), "+v"(qy[d][1]), "+v"(qa[d][0]), "+v"(qa[d][1]), "+v"(qa[d][2])
                         : "n"(3 * LPS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qy[d][0]), "+v"(qa[d][0]), "+v"(qa[d][1]), "+v"(qa[d][2])
                         : "n"(3 * LPS) : "memory");
        if constexpr (GM == 2) {
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("" : "+v"(qs[d][k]));
        }
    };
    auto pin_accumulators = [&]() {
        // every accumulator in the accumulation registers at the loop's entry and back edge: in the narrow forms hipcc
        // kept some of them in vector registers there and moved up to 80 dwords in and out every iteration
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int t = 0; t < CT; ++t) asm volatile("" : "+a"(acc[rt][t]));
        if constexpr (GM == 2) {
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("" : "+a"(gsp[k]));
        } else {
#pragma unroll
            for (int gr = 0; gr < NG; ++gr)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) asm volatile("" : "+a"(gacc[gr][ct]));
        }
    };
    pin_accumulators();
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    for (int u = u_lo; u < u_hi; u += D) {  // the slice is whole 16-sample tiles: a multiple of D steps
#pragma unroll
        for (int d = 0; d < D; ++d) {
            landed(d);
            __builtin_amdgcn_sched_barrier(0);
            double a[3], y[CT];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) a[rt] = __builtin_bit_cast(double, qa[d][rt]);
            y[0] = __builtin_bit_cast(double, u2_t{qy[d][0].x, qy[d][0].y});
            if constexpr (CT >= 2) y[1] = __builtin_bit_cast(double, u2_t{qy[d][0].z, qy[d][0].w});
            if constexpr (CT == 4) {
                y[2] = __builtin_bit_cast(double, u2_t{qy[d][1].x, qy[d][1].y});
                y[3] = __builtin_bit_cast(double, u2_t{qy[d][1].z, qy[d][1].w});
            }
            double asp[NS == 0 ? 1 : NS];
#pragma unroll
            for (int k = 0; k < (NS == 0 ? 1 : NS); ++k) asp[k] = GM == 2 ? __builtin_bit_cast(double, qs[d][k]) : 0.0;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                yy[t] = fma(y[t], y[t], yy[t]);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
                    acc[rt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], y[t], acc[rt][t], 0, 0, 0);
            }
            if constexpr (GM == 2) {
                // special rows x the wave's column tiles (NW = 4: tile min(wave, 2); NW = 2: wave 0 tiles 0 and 1,
                // wave 1 tile 2 twice; NW = 1: all three) -- straight-line code, as for the full matrix below
                if constexpr (NW == 4) {
                    const double bw_ = wv == 0 ? a[0] : (wv == 1 ? a[1] : a[2]);
                    gsp[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(asp[0], bw_, gsp[0], 0, 0, 0);
                } else if constexpr (NW == 2) {
                    const double b0 = wv == 0 ? a[0] : a[2], b1 = wv == 0 ? a[1] : a[2];
                    gsp[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(asp[0], b0, gsp[0], 0, 0, 0);
                    gsp[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(asp[1], b1, gsp[1], 0, 0, 0);
                } else {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
                        gsp[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(asp[ct], a[ct], gsp[ct], 0, 0, 0);
                }
            } else if constexpr (NW == 4) {
                // W'W row-tile min(wave, 2): straight-line (a wave-uniform branch here made the
                // compiler shuttle the Gram accumulators between register files every step)
                const double aw = wv == 0 ? a[0] : (wv == 1 ? a[1] : a[2]);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    gacc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, a[ct], gacc[0][ct], 0, 0, 0);
            } else if constexpr (NW == 2) {
                const double aw0 = wv == 0 ? a[0] : a[2], aw1 = wv == 0 ? a[1] : a[2];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    gacc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw0, a[ct], gacc[0][ct], 0, 0, 0);
                    gacc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw1, a[ct], gacc[1][ct], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int gr = 0; gr < 3; ++gr)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
                        gacc[gr][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[gr], a[ct], gacc[gr][ct], 0, 0, 0);
            }
            // the requests of step u + d + D (past the end of the data: zeros) go out AFTER this step's products have
            // been issued, into the registers those have just read
            __builtin_amdgcn_sched_barrier(0);
            fetch(d);
            __builtin_amdgcn_sched_barrier(0);
        }
        pin_accumulators();
    }
    // The requests past the slice's end are still in flight here and the compiler does not know: it would compute the
    // epilogue's addresses in their destination registers (dead, to its knowledge) and a late return would overwrite
    // them (seen: a store through such an address faulted).  The ties BEHIND the wait keep every destination register
    // allocated until the queue has drained.  (Requesting nothing in the last iteration instead makes the operands
    // phi values, and the copies the compiler then places on the loop's back edge read registers whose data has not
    // landed.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int k = 0; k < NYL; ++k) asm volatile("" : "+v"(qy[d][k]));
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) asm volatile("" : "+v"(qa[d][rt]));
        if constexpr (GM == 2) {
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("" : "+v"(qs[d][k]));
        }
    }
    // y'y: the four sample lanes of a column hold partial sums
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        yy[t] += __shfl_xor(yy[t], 16);
        yy[t] += __shfl_xor(yy[t], 32);
    }
    // D[row (lane>>4) + 4v][column lane&15] -> stacked row 16*rt + row, data column 64*wave + 4c + t
    auto block_of = [&](int pc) -> double * {
        return part + ((static_cast<int64_t>(pc) * nz + zb) * nsplit + sl) * static_cast<int64_t>(kpc + 1) * kNCol;
    };
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 16 * rt + sj + 4 * v;
            // candidate and row within it: GM = 2 packed (global row g0 + r = K pc + k), GM = 1 padded to kpc rows
            const int pc = GM == 2 ? (g0 + r) / K : pb * ncb + r / kpc;
            const int k = GM == 2 ? g0 + r - pc * K : r % kpc;
            if (pc >= n_periods) continue;
            double *row = block_of(pc) + static_cast<int64_t>(k) * kNCol;
#pragma unroll
            for (int t = 0; t < CT; ++t) row[col0 + t] = acc[rt][t][v];
        }
    }
    if (GM == 2 && wv < 3) {
        // D[special sj + 4v][column 16*ct + c of the stack]: kept where the column belongs to the special row's
        // candidate, in the cell the full matrix has for it (rows 0, K-2, K-1, column k of the candidate's W'W)
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) {
            if (NW == 2 && wv == 1 && kk == 1) continue;  // (the repeated tile)
            const int gl = g0 + 16 * tile_of(kk), g = gl + c;
            const int pg = g / K, k = g - pg * K;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = sj + 4 * v, ci = i / 3, t = i - 3 * ci;
                const int pc = gl / K + ci;
                if (pc != pg || pc >= n_periods) continue;
                const int row = t == 0 ? 0 : K - 3 + t;
                block_of(pc)[static_cast<int64_t>(row) * kNCol + kYCols + k] = gsp[kk][v];
            }
        }
    }
    if (GM == 1 && wv < 3) {
#pragma unroll
        for (int gr = 0; gr < NG; ++gr) {
            if (NW == 2 && wv == 1 && gr == 1) continue;  // (the repeated tile)
            const int rt = NW == 4 ? wv : (NW == 2 ? (wv == 0 ? gr : 2) : gr);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * rt + sj + 4 * v;
                const int pc = pb * ncb + r / kpc;
                if (pc >= n_periods) continue;
                double *row = block_of(pc) + static_cast<int64_t>(r % kpc) * kNCol + kYCols;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) row[16 * ct + c] = gacc[gr][ct][v];
            }
        }
    }
    if (sj == 0) {
        // (GM = 2: every candidate with rows in this stack -- one that straddles gets the same sums from both)
        const int p_lo = GM == 2 ? g0 / K : pb * ncb, p_hi = GM == 2 ? (g0 + KS - 1) / K : pb * ncb + ncb - 1;
        for (int pc = p_lo; pc <= p_hi; ++pc) {
            if (pc >= n_periods) break;
            double *row = block_of(pc) + static_cast<int64_t>(kpc) * kNCol;
#pragma unroll
            for (int t = 0; t < CT; ++t) row[col0 + t] = yy[t];
        }
    }
}
template <int NW, int CT = 4, int GM = 1>
__global__ void __launch_bounds__(256) fit_accum_mfma_kernel(const double *Y, int64_t ldy, const double *Ws,
                                                              int n_pad, int n_idx, int n_chans, int n_periods,
                                                              int kpc, int nsplit, double *part, int kreal) {
    fit_accum_mfma_body<NW, CT, GM>(Y, ldy, Ws, n_pad, n_idx, n_chans, n_periods, kpc, nsplit, part, this_block(), kreal);
}

// FUSED form of the packed / special-rows Gram kernel (round 3, 256-channel blocks): the design rows are not read from
// memory -- the workgroup computes them itself, 64 samples at a time, into an LDS tile (the arithmetic of
// fit_trig_packed_body statement for statement: one sincos per candidate and sample, the recurrence for the harmonics),
// while the matrix cores work on the previous tile.  The MFMA operands are the same values in the same order as in
// fit_accum_mfma_body<4, 4, 2>, so the partial blocks are its bits; what goes away is the design-matrix kernel (its
// 4.4 GB of writes at the 10 044-candidate grid, 0.7 ms; 20 us + a launch in front of every optimiser batch) and the
// Gram kernel's operand traffic for it.
//  * LDS tile: row = sample (64), FS = 67 doubles per row: the stack's 48 design rows, then rows 0, K-2, K-1 of each of
//    the <= 6 candidates with rows in the stack (the special A operand: a straddling candidate's rows are computed
//    here rather than fetched from the neighbouring stack), then one zero.  Two tiles (68.6 KB): wave q computes
//    candidates q, q + 4 of tile i + 1 right after the barrier that opens tile i.
//  * One barrier per tile (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would also drain the data requests).
//  * The sample indices of tile i + 2 are requested (one buffer load per lane) when tile i opens, in the same counted
//    queue as the data quads; every data wait keeps the 3-steps-younger rule, which with the index request in the queue
//    waits for at most one request more than needed.
constexpr int kFusedRow = 67, kFusedTile = 64 * kFusedRow;                    // doubles
constexpr int kFusedLdsBytes = (2 * kFusedTile + 4 * kFusedRow) * 8;          // (+ the rows the last look-ahead read touches)
__device__ __forceinline__ void fit_accum_fused_body(const double *Y, int64_t ldy, const int64_t *idx, const double *periods,
                                                     int n_pad, int n_idx, int n_chans, int n_periods, int kpc, int nsplit,
                                                     double *part, const Blk blk, int K) {
    extern __shared__ double fused_lds[];
    constexpr int KS = 48, D = 4, CT = 4, FS = kFusedRow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, sj = lane >> 4;
    const int sl = blk.x, pb = blk.y, zb = blk.z, nz = blk.nz;
    const int tiles = n_pad / kTJ;
    const int per = (tiles + nsplit - 1) / nsplit;
    const int u_lo = sl * per * 4;  // steps of 4 samples
    const int u_hi = min(tiles, sl * per + per) * 4;
    const int col0 = 64 * wv + 4 * c;
    const bool col_ok = zb * kYCols + col0 < n_chans;
    const unsigned step_y = static_cast<unsigned>(4 * ldy * 8);
    unsigned vy = col_ok ? static_cast<unsigned>(((static_cast<int64_t>(u_lo) * 4 + sj) * ldy + col0) * 8) : 0x80000000u;
    const int g0 = KS * pb, p0 = g0 / K;
    const int n_cand = (KS + K - 2) / K + 1;  // packed_trig_waves(K)
    // the special operand of this wave's column tile (tile min(wave, 2)): lane c reads special row c % 3 of the
    // (c / 3)-th candidate that owns columns of the tile, or the zero
    const int gl = g0 + 16 * min(wv, 2);
    int spcol;
    {
        const int ci = c / 3, t = c - 3 * ci, pc = gl / K + ci;
        const bool own = pc <= (gl + 15) / K && pc < n_periods;
        spcol = own ? KS + 3 * (pc - p0) + t : FS - 1;
    }
    for (int e = tid; e < 2 * 64 + 4; e += 256) fused_lds[e * FS + FS - 1] = 0.0;
    // the <= 2 candidates whose rows this wave computes
    // (two scalars, not an array: indexed by the loop below it would live in scratch, and every scratch read drains the
    // request queue)
    const double w0a = p0 + wv < n_periods ? 6.283185307179586 / periods[p0 + wv] : 0.0;  // (2 * np.pi / period)
    const double w0b = wv + 4 < n_cand && p0 + wv + 4 < n_periods ? 6.283185307179586 / periods[p0 + wv + 4] : 0.0;

    d4_t acc[3][CT], gsp;
    double yy[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) yy[t] = 0.0;
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = d4_t{0.0, 0.0, 0.0, 0.0};
    gsp = d4_t{0.0, 0.0, 0.0, 0.0};

    constexpr int LPS = 2;  // requests per step: the two data quads
    auto desc_of = [](const void *base, int bytes) -> u4_t {
        const unsigned long long p = reinterpret_cast<unsigned long long>(base);
        return u4_t{static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(p))),
                    static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(p >> 32))) & 0xffffu,
                    static_cast<unsigned>(__builtin_amdgcn_readfirstlane(bytes)), 0x00020000u};
    };
    const u4_t dy = desc_of(Y + static_cast<int64_t>(zb) * kYCols,
                            static_cast<int>((static_cast<int64_t>(n_idx) * ldy - static_cast<int64_t>(zb) * kYCols) * 8));
    const u4_t di = desc_of(idx, n_idx * 8);  // (beyond the last index: zeros, and the rows are zeros then)
    u4_t qy[D][2];
    u2_t qi;
    unsigned vi = static_cast<unsigned>((u_lo * 4 + lane) * 8);  // this lane's sample of the tile whose indices come next
    auto fetch = [&](int d) {
        asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:16"
                     : "=&v"(qy[d][0]), "=&v"(qy[d][1]) : "v"(vy), "s"(dy) : "memory");
        if (col_ok) vy += step_y;
    };
    auto fetch_indices = [&](u2_t &into) {
        asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(into) : "v"(vi), "s"(di) : "memory");
        vi += 64 * 8;
    };
    auto landed = [&](int d) {
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(qy[d][0]), "+v"(qy[d][1]) : "n"(3 * LPS) : "memory");
    };
    auto pin_accumulators = [&]() {
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int t = 0; t < CT; ++t) asm volatile("" : "+a"(acc[rt][t]));
        asm volatile("" : "+a"(gsp));
    };
    // design rows of the 64 samples from `j0` on into `tile` (this wave's candidates; `qi` holds the lane's index)
    auto design_rows = [&](int j0, double *tile, const u2_t &index) {
        const int j = j0 + lane;
        const unsigned index_lo = index.x, index_hi = index.y;  // (components through scalars: see parrm_filter_comb.hip)
        const int64_t at = static_cast<int64_t>((static_cast<unsigned long long>(index_hi) << 32) | index_lo);
        auto candidate = [&](int q, double w0) {
            const int p = p0 + q;
            const int k_lo = max(g0 - p * K, 0), k_hi = min(g0 + KS - p * K, K);  // its rows [k_lo, k_hi) lie in the stack
            if (k_lo >= k_hi) return;                     // (the last of the <= n_cand may begin beyond the stack)
            double *t = tile + lane * FS + (p * K - g0);  // t[k] = row k of the candidate (k_lo <= k < k_hi)
            double *sp = tile + lane * FS + KS + 3 * q;   // its rows 0, K-2, K-1
            if (j >= n_idx || p >= n_periods) {
                for (int k = k_lo; k < k_hi; ++k) t[k] = 0.0;
                sp[0] = 0.0;
                sp[1] = 0.0;
                sp[2] = 0.0;
            } else {
                const double ang = static_cast<double>(at + 1) * w0;  // (indices + 1) * ...
                if (k_lo == 0) t[0] = 1.0;
                sp[0] = 1.0;
                double s1, c1;
                sincos(ang, &s1, &c1);
                double sk = s1, ck = c1;
                auto next_harmonic = [&]() {
                    const double sn = fma(sk, c1, ck * s1);
                    const double cn = fma(ck, c1, -(sk * s1));
                    sk = sn;
                    ck = cn;
                };
                // harmonic k owns rows 2k-1 (sin) and 2k (cos): below the stack's window [k_lo, k_hi) nothing is stored, in
                // its first and last harmonic perhaps one of the two, between them both, beyond it nothing -- loops without
                // per-row tests (those cost more than the recurrence: ~25 scalar instructions and three branches a harmonic)
                const int bw = (K - 1) / 2;
                const int hf = (max(k_lo, 1) + 1) / 2, hl = k_hi / 2;  // first / last harmonic with a row in the window
                int k = 1;                                             // (sk, ck) = harmonic k
                for (; k < hf; ++k) next_harmonic();
                if (hf <= hl) {
                    if (2 * k - 1 >= k_lo) t[2 * k - 1] = sk;
                    if (2 * k < k_hi) t[2 * k] = ck;
                    if (hl > hf) {
                        next_harmonic();
                        ++k;
                        double *row = t + 2 * k - 1;
                        for (; k < hl; ++k) {
                            row[0] = sk;
                            row[1] = ck;
                            row += 2;
                            next_harmonic();
                        }
                        t[2 * k - 1] = sk;
                        if (2 * k < k_hi) t[2 * k] = ck;
                    }
                }
                for (; k < bw; ++k) next_harmonic();
                sp[1] = sk;
                sp[2] = ck;
            }
        };
        if (wv < n_cand) candidate(wv, w0a);
        if (wv + 4 < n_cand) candidate(wv + 4, w0b);
    };

    // prologue: the first two tiles' indices, four steps' data (in flight while the first tile's rows are computed)
    u2_t qi0;
    fetch_indices(qi0);
    fetch_indices(qi);
    pin_accumulators();
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(qi0), "+v"(qi) : "n"(4 * LPS) : "memory");
    design_rows(u_lo * 4, fused_lds, qi0);
    pin_accumulators();
    const int lane_a = sj * FS + c, lane_s = sj * FS + spcol;
    const double *cur = fused_lds + kFusedTile;  // (swapped when the first tile opens)
    double a_nx[3], s_nx;
    for (int u = u_lo; u < u_hi; u += D) {  // four steps = 16 samples per iteration; a tile every fourth
        const int g = ((u - u_lo) >> 2) & 3;
        if (g == 0) {
            // everyone has left the previous tile (its buffer is free) and has written this one
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            double *nxt = const_cast<double *>(cur);
            cur = cur == fused_lds ? fused_lds + kFusedTile : fused_lds;
            // (the index request is older than the four steps' requests in flight)
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qi) : "n"(4 * LPS) : "memory");
            design_rows(u * 4 + 64, nxt, qi);
            fetch_indices(qi);
            pin_accumulators();
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) a_nx[rt] = cur[lane_a + 16 * rt];
            s_nx = cur[lane_s];
        }
        const double *row = cur + (16 * g) * FS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            landed(d);
            __builtin_amdgcn_sched_barrier(0);
            double a[3], y[CT];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) a[rt] = a_nx[rt];
            const double asp = s_nx;
            // the next step's design rows (after the tile's last step: rows nobody uses), requested BEFORE this step's
            // products, into registers of their own (left to itself hipcc reuses the registers and requests them after
            // the products).  In the first step of an iteration they go out behind the first column's products instead:
            // at the loop's head hipcc waits for lgkmcnt(0) before the first product, which with the look-ahead
            // already out would be a wait for reads issued a few cycles earlier.
            auto look_ahead = [&]() {
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) a_nx[rt] = row[(4 * d + 4) * FS + lane_a + 16 * rt];
                s_nx = row[(4 * d + 4) * FS + lane_s];
            };
            if (d != 0) {
                look_ahead();
                __builtin_amdgcn_sched_barrier(0);
            }
            y[0] = __builtin_bit_cast(double, u2_t{qy[d][0].x, qy[d][0].y});
            y[1] = __builtin_bit_cast(double, u2_t{qy[d][0].z, qy[d][0].w});
            y[2] = __builtin_bit_cast(double, u2_t{qy[d][1].x, qy[d][1].y});
            y[3] = __builtin_bit_cast(double, u2_t{qy[d][1].z, qy[d][1].w});
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                yy[t] = fma(y[t], y[t], yy[t]);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
                    acc[rt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], y[t], acc[rt][t], 0, 0, 0);
                if (d == 0 && t == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    look_ahead();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const double bw_ = wv == 0 ? a[0] : (wv == 1 ? a[1] : a[2]);
            gsp = __builtin_amdgcn_mfma_f64_16x16x4f64(asp, bw_, gsp, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(d);
            __builtin_amdgcn_sched_barrier(0);
        }
        pin_accumulators();
    }
    // (requests past the slice's end are still in flight: see fit_accum_mfma_body)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; ++d) {
        asm volatile("" : "+v"(qy[d][0]));
        asm volatile("" : "+v"(qy[d][1]));
    }
    asm volatile("" : "+v"(qi));
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        yy[t] += __shfl_xor(yy[t], 16);
        yy[t] += __shfl_xor(yy[t], 32);
    }
    auto block_of = [&](int pc) -> double * {
        return part + ((static_cast<int64_t>(pc) * nz + zb) * nsplit + sl) * static_cast<int64_t>(kpc + 1) * kNCol;
    };
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 16 * rt + sj + 4 * v;
            const int pc = (g0 + r) / K, k = g0 + r - pc * K;
            if (pc >= n_periods) continue;
            double *out = block_of(pc) + static_cast<int64_t>(k) * kNCol;
#pragma unroll
            for (int t = 0; t < CT; ++t) out[col0 + t] = acc[rt][t][v];
        }
    }
    if (wv < 3) {
        const int g = gl + c, pg = g / K, k = g - pg * K;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = sj + 4 * v, ci = i / 3, t = i - 3 * ci;
            const int pc = gl / K + ci;
            if (pc != pg || pc >= n_periods) continue;
            const int out_row = t == 0 ? 0 : K - 3 + t;
            block_of(pc)[static_cast<int64_t>(out_row) * kNCol + kYCols + k] = gsp[v];
        }
    }
    if (sj == 0) {
        const int p_hi = (g0 + KS - 1) / K;
        for (int pc = p0; pc <= p_hi; ++pc) {
            if (pc >= n_periods) break;
            double *out = block_of(pc) + static_cast<int64_t>(kpc) * kNCol;
#pragma unroll
            for (int t = 0; t < CT; ++t) out[col0 + t] = yy[t];
        }
    }
}
__global__ void __launch_bounds__(256) fit_accum_fused_kernel(const SmallPeriods, const double *Y, int64_t ldy,
                                                               const int64_t *idx, const double *periods, int n_pad,
                                                               int n_idx, int n_chans, int n_periods, int kpc, int nsplit,
                                                               double *part, int kreal, int by_value) {
    fit_accum_fused_body(Y, ldy, idx, by_value ? small_periods_in_kernarg() : periods, n_pad, n_idx, n_chans, n_periods, kpc,
                         nsplit, part, this_block(), kreal);
}

// red[(p*nz+zb)][e] = sum_s part[(p*nz+zb)][s][e], s ascending (deterministic)
// (data columns beyond the recording's channels are never read by the solvers: skipped)
// (ksp = K when the Gram kernel left only rows 0, K-2, K-1 of W'W -- its GM = 2 form --, else 0)
__device__ __forceinline__ void fit_reduce_body(const double *part, int nsplit, int64_t elems, int nz, int n_chans,
                                                double *red, const Blk blk, int ksp) {
    const int64_t e = static_cast<int64_t>(blk.x) * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    const int64_t pz = blk.y;
    const int col = static_cast<int>(e % kNCol), zb = static_cast<int>(pz % nz);
    if (col < kYCols && zb * kYCols + col >= n_chans) return;
    if (ksp > 0 && col >= kYCols) {
        const int row = static_cast<int>(e / kNCol);
        if (row != 0 && row != ksp - 2 && row != ksp - 1) return;
    }
    const double *src = part + pz * nsplit * elems + e;
    // the loads of eight slices are in flight together (a plain loop waited for each in turn: 18 us per
    // optimiser batch at 56 slices); the sum is still taken in slice order
    double a = 0.0;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[static_cast<int64_t>(s + k) * elems];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += v[k];
    }
    for (; s < nsplit; ++s) a += src[static_cast<int64_t>(s) * elems];
    red[pz * elems + e] = a;
}
__global__ void __launch_bounds__(256) fit_reduce_kernel(const double *part, int nsplit, int64_t elems, int nz,
                                                          int n_chans, double *red, int ksp) {
    fit_reduce_body(part, nsplit, elems, nz, n_chans, red, this_block(), ksp);
}

// One workgroup per candidate period: LU(G) with partial pivoting, solve for every channel,
// err = mean_c[(y'y - 2 b'R + b'G b)/n + regu . b^2].
__global__ void __launch_bounds__(256) fit_solve_kernel(const double *red, int n_idx, int n_chans, int bw,
                                                         int KP, int nz, double lambda, double *err) {
    const int goff = (blockIdx.x % (48 / KP)) * KP;  // this candidate's block of the stacked W columns
    constexpr int KS = 49;   // padded row stride of the K x K matrices
    constexpr int RS = 257;  // padded row stride of the K x 256 right-hand sides
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *A = reinterpret_cast<double *>(lds_raw);  // [K][KS] LU factors
    double *G0 = A + 48 * KS;                         // [K][KS] original Gram matrix
    double *Rl = G0 + 48 * KS;                        // [K][RS] rhs -> beta
    __shared__ int piv[48];
    __shared__ int singular;
    __shared__ double wsum[4];

    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    const int K = 2 * bw + 1;
    const int64_t elems = static_cast<int64_t>(KP + 1) * kNCol;
    const double *base = red + static_cast<int64_t>(p) * nz * elems;

    for (int e = tid; e < K * K; e += 256) {
        const int r = e / K, c = e % K;
        const double g = base[r * kNCol + kYCols + goff + c];
        A[r * KS + c] = g;
        G0[r * KS + c] = g;
    }
    if (tid == 0) singular = 0;
    __syncthreads();

    for (int col = 0; col < K; ++col) {
        if (tid < 64) {  // idamax over rows col..K-1 (first maximum wins, like LAPACK)
            const int r = col + tid;
            double v = r < K ? fabs(A[r * KS + col]) : -1.0;
            if (v != v) v = 1e308;  // a NaN column: let it through as the pivot, like a NaN compare chain
            int best = r;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(v, off, 64);
                const int ob = __shfl_xor(best, off, 64);
                if (ov > v || (ov == v && ob < best)) {
                    v = ov;
                    best = ob;
                }
            }
            if (tid == 0) piv[col] = best;
        }
        __syncthreads();
        const int pr = piv[col];
        if (pr != col && tid < K) {
            const double tmp = A[col * KS + tid];
            A[col * KS + tid] = A[pr * KS + tid];
            A[pr * KS + tid] = tmp;
        }
        __syncthreads();
        const double pv = A[col * KS + col];
        if (pv == 0.0) {  // exactly singular: numpy raises LinAlgError -> inf (:627-628)
            if (tid == 0) singular = 1;
            break;  // pv is uniform across the workgroup
        }
        if (tid > col && tid < K) A[tid * KS + col] /= pv;
        __syncthreads();
        const int m = K - col - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = col + 1 + e / m, j = col + 1 + e % m;
            A[i * KS + j] -= A[i * KS + col] * A[col * KS + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (singular) {
        if (tid == 0) err[p] = INFINITY;
        return;
    }

    const double ksum = static_cast<double>(K * (K + 1) / 2);
    double total = 0.0;
    for (int zb = 0; zb < nz; ++zb) {
        const double *rz = base + zb * elems;
        const int ncol = min(kYCols, n_chans - zb * kYCols);
        for (int e = tid; e < K * kYCols; e += 256) {
            const int r = e / kYCols, c = e % kYCols;
            Rl[r * RS + c] = c < ncol ? rz[r * kNCol + c] : 0.0;
        }
        __syncthreads();
        double e_c = 0.0;
        if (tid < ncol) {
            double *b = Rl + tid;
            for (int i = 0; i < K; ++i) {  // apply the row interchanges
                const int pr = piv[i];
                if (pr != i) {
                    const double tmp = b[i * RS];
                    b[i * RS] = b[pr * RS];
                    b[pr * RS] = tmp;
                }
            }
            for (int i = 1; i < K; ++i) {  // L z = P r
                double v = b[i * RS];
                for (int j = 0; j < i; ++j) v -= A[i * KS + j] * b[j * RS];
                b[i * RS] = v;
            }
            for (int i = K - 1; i >= 0; --i) {  // U beta = z
                double v = b[i * RS];
                for (int j = i + 1; j < K; ++j) v -= A[i * KS + j] * b[j * RS];
                b[i * RS] = v / A[i * KS + i];
            }
            double t1 = 0.0, t2 = 0.0, reg = 0.0;
            for (int i = 0; i < K; ++i) {
                const double bi = b[i * RS];
                t1 = fma(bi, rz[i * kNCol + tid], t1);
                double gb = 0.0;
                for (int j = 0; j < K; ++j) gb = fma(G0[i * KS + j], b[j * RS], gb);
                t2 = fma(bi, gb, t2);
                reg += (lambda * static_cast<double>(i + 1) / ksum) * (bi * bi);  // :585-586,:595
            }
            const double yy = rz[KP * kNCol + tid];
            e_c = (yy - 2.0 * t1 + t2) / static_cast<double>(n_idx) + reg;
        }
        const double v = parrm::wave_sum(e_c);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        total += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        __syncthreads();
    }
    if (tid == 0) err[p] = total / static_cast<double>(n_chans);  // :597
}

// Fast solve for the bandwidths the schedule uses (K = 11, 21, 41 known at compile time):
//   * the LU factorisation (same algorithm as above: partial pivoting, first maximum wins, exact
//     zero pivot -> +inf) runs in ONE wave with wave-synchronous LDS traffic instead of four
//     workgroup barriers per column;
//   * every thread then solves one channel with its right-hand side in REGISTERS (fully unrolled
//     substitution; the factors come from LDS as broadcast reads), and evaluates the quadratic form.
// Measured on the K = 41 polish batches: 210 us -> see profiles.
// max over a DPP-shuffled copy of a 64-bit key (rows not in ROW_MASK keep their value)
template <int CTRL, int ROW_MASK>
__device__ inline unsigned long long dpp_max_u64(unsigned long long v) {
    const int lo = static_cast<int>(v), hi = static_cast<int>(v >> 32);
    const unsigned olo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false));
    const unsigned ohi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false));
    const unsigned long long o = (static_cast<unsigned long long>(ohi) << 32) | olo;
    return o > v ? o : v;
}

// Elimination in the natural order, one matrix row per lane, the row in registers (the pivot row is broadcast with
// v_readlane at a compile-time lane: no pivot search, no lane index in an SGPR).  Used for the matrix rebuilt from
// its three rows: W'W is symmetric positive definite, the natural order is backward stable for it (growth factor 1)
// and LAPACK's search picks the diagonal anyway (the errors of the bench shapes come out bit-identical to the pivoting
// form's); multipliers through the pivot's reciprocal, as dgetf2 scales its column.  In-kernel: 22 -> 12 us of a
// K = 41 batch.  A pivot <= tiny marks the system singular.
template <int K>
__device__ __forceinline__ bool lu_natural_order(double (&arow)[K], int lane, double tiny) {
    bool sing = false;
#pragma unroll
    for (int col = 0; col < K; ++col) {
        double prow[K];
#pragma unroll
        for (int j = col; j < K; ++j) {
            const long long bits = __double_as_longlong(arow[j]);
            const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), col);
            const int hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), col);
            prow[j] = __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
        }
        if (fabs(prow[col]) <= tiny) {
            sing = true;
            break;
        }
        const double rinv = 1.0 / prow[col];
        if (lane > col) {
            const double l = arow[col] * rinv;
            arow[col] = l;
#pragma unroll
            for (int j = col + 1; j < K; ++j) arow[j] -= l * prow[j];
        }
    }
    return sing;
}

// NW = 4: one workgroup of four waves per candidate (grids).  NW = 1, the form of the optimiser's small batches:
// one WAVE per (candidate, 64-channel quarter of a 256-channel block), blockIdx.y = 4*zb + quarter.  Every wave
// factorises the candidate's Gram matrix for itself -- idle CUs are free in a batch of a few candidates -- and
// has an LDS of its own for the substitutions, where the four waves of the NW = 4 form contend for one (the
// substitution phase is LDS-bound: 840 broadcast reads per wave).  It leaves its wave sum in a spare cell of the
// candidate's block (row KP, columns kYCols + quarter: the y'y row has no Gram part) and fit_finish_kernel adds
// the quarters in the order the NW = 4 form adds its waves: same bits, 58 -> 35 us per K = 41 batch.
//
// `special` != 0: the block holds rows 0, K-2 and K-1 of W'W only (the Gram kernel's GM = 2 form) and the matrix is
// rebuilt from the moments C_m = sum cos(m a), S_m = sum sin(m a), m = 0 ... 2 bw, which those rows determine:
//   m <= bw:  S_m = G[0][2m-1], C_m = G[0][2m]  (C_0 = G[0][0] = n);
//   m = bw+i: S_m = G[sin_bw][cos_i] + G[cos_bw][sin_i],  C_m = G[cos_bw][cos_i] - G[sin_bw][sin_i];
//   G[sin_k][sin_l] = (C_|k-l| - C_k+l)/2,  G[cos_k][cos_l] = (C_|k-l| + C_k+l)/2,  G[sin_k][cos_l] = (S_k+l + S_k-l)/2
// (S_-m = -S_m).  Rows are ordered 1, sin(a), cos(a), sin(2a), ... (fit_trig_stacked_body).
template <int K, int NT>
__device__ __forceinline__ void gram_from_special_rows(const double *base, int goff, double *A, int KS, int tid) {
    constexpr int BW = (K - 1) / 2;
    __shared__ double rows3[3][K + 1];
    __shared__ double Cm[2 * BW + 1], Sm[2 * BW + 1];
    for (int e = tid; e < 3 * K; e += NT) {
        const int t = e / K, c = e - t * K;
        rows3[t][c] = base[static_cast<int64_t>(t == 0 ? 0 : K - 3 + t) * kNCol + kYCols + goff + c];
    }
    __syncthreads();
    for (int m = tid; m <= 2 * BW; m += NT) {
        double sm, cm;
        if (m == 0) {
            sm = 0.0;
            cm = rows3[0][0];
        } else if (m <= BW) {
            sm = rows3[0][2 * m - 1];
            cm = rows3[0][2 * m];
        } else {
            const int i = m - BW;
            sm = rows3[1][2 * i] + rows3[2][2 * i - 1];
            cm = rows3[2][2 * i] - rows3[1][2 * i - 1];
        }
        Sm[m] = sm;
        Cm[m] = cm;
    }
    __syncthreads();
    for (int e = tid; e < K * K; e += NT) {
        const int r = e / K, c = e - r * K;
        double v;
        if (r == 0 || c == 0) {
            v = rows3[0][r + c];
        } else {
            const int k = (r + 1) >> 1, l = (c + 1) >> 1;  // harmonics; odd index = sine row
            const bool rs = r & 1, cs = c & 1;
            const int d = k > l ? k - l : l - k;
            if (rs == cs) {
                v = rs ? 0.5 * (Cm[d] - Cm[k + l]) : 0.5 * (Cm[d] + Cm[k + l]);
            } else {
                // sin(x) cos(y) = (sin(x + y) + sin(x - y)) / 2, x the sine's harmonic
                const int x = rs ? k : l, y = rs ? l : k;
                const double sd = x >= y ? Sm[x - y] : -Sm[y - x];
                v = 0.5 * (Sm[k + l] + sd);
            }
        }
        A[r * KS + c] = v;
    }
}

template <int K, int NW>
__device__ __forceinline__ void fit_solve_fast_body(const double *red, int n_idx, int n_chans, int KP, int nz, double lambda,
                                                    double *err, const Blk blk, int special) {
    constexpr int KS = 49;  // odd row stride: lanes-as-rows accesses in the LU stay bank-conflict free
    constexpr int KE = (K + 2) & ~1;  // even row stride of the substitution copy: 16-byte aligned rows
    __shared__ double A[K * KS];    // Gram matrix in, read one row per lane by the LU
    __shared__ __attribute__((aligned(16))) double LU[K * KE];  // packed factors, rows in pivot order
    __shared__ int perm[64];        // perm[i] = original row that ended up as row i
    __shared__ int singular;
    __shared__ double wsum[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = blk.x;
    const int goff = special ? 0 : (p % (48 / KP)) * KP;  // (the packed Gram form leaves column k at kYCols + k)
    const int64_t elems = static_cast<int64_t>(KP + 1) * kNCol;
    const double *base = red + static_cast<int64_t>(p) * nz * elems;
    const int quarter = NW == 4 ? (tid >> 6) : static_cast<int>(blk.y & 3);  // 64-channel quarter of a block
    const int ch = quarter * 64 + lane;                                             // channel within the block

    if (special) {
        gram_from_special_rows<K, 64 * NW>(base, goff, A, KS, tid);
    } else {
        for (int e = tid; e < K * K; e += 64 * NW) {
            const int r = e / K, c = e % K;
            A[r * KS + c] = base[r * kNCol + kYCols + goff + c];
        }
    }
    if (tid < 64) perm[tid] = tid;
    if (tid == 0) singular = 0;
    __syncthreads();

    if (tid < 64) {
        // wave 0: LU with partial pivoting, one matrix row per lane, the row in REGISTERS.  Rows are
        // never moved: a lane that has served as pivot row simply stops taking part.  Step `col`
        // picks the pivot lane (largest |a[col]|, lowest original row on ties = LAPACK's first
        // maximum), broadcasts that lane's row with v_readlane and eliminates.  Lane i's
        // multipliers stay in a[0..rank_i) and its U entries in a[rank_i..K), so writing the row to
        // LDS row rank_i gives exactly the packed LU of the row-permuted matrix.
        double arow[K];
#pragma unroll
        for (int j = 0; j < K; ++j) arow[j] = lane < K ? A[lane * KS + j] : 0.0;
        // Singular = LAPACK's exact zero pivot.  A matrix rebuilt from its moments carries ~4 eps n of rounding per
        // entry where the full product has exact duplicates (a period so long that every cosine rounds to 1: columns
        // that are EQUAL in W'W differ in the last bits here), so there a pivot within that noise of zero -- below
        // 1e-12 n, n = G[0][0]; the pivots of a sound system are ~n/2 -- counts as the zero it stands for.
        const double tiny = special ? 1e-12 * fabs(A[0]) : 0.0;
        bool done = lane >= K;
        int rank = -1;
        bool sing = false;
        if (special) {
            rank = lane;
            sing = lu_natural_order<K>(arow, lane, tiny);
        } else
#pragma unroll
        for (int col = 0; col < K; ++col) {
            double v = fabs(arow[col]);
            if (v != v) v = 1e308;  // a NaN: let it through as the pivot
            // arg-max as ONE unsigned 64-bit max over the wave: a non-negative double orders like its
            // bit pattern; its low 6 bits are replaced by 63 - lane (ties -> lowest row; values that
            // differ only below 2^-46 relative count as ties -- either is as good a pivot), +64 keeps
            // every candidate above the 0 of the lanes that have already served.  Row-wise DPP
            // reduction (quad swaps, half-mirror, mirror, two row broadcasts: rocPRIM's gfx9 scheme)
            // instead of six rounds of three ds_bpermutes.
            unsigned long long key =
                done ? 0ull : ((static_cast<unsigned long long>(__double_as_longlong(v)) & ~63ull) | (63u - lane)) + 64ull;
            key = dpp_max_u64<0xB1, 0xf>(key);   // quad_perm [1,0,3,2]
            key = dpp_max_u64<0x4E, 0xf>(key);   // quad_perm [2,3,0,1]
            key = dpp_max_u64<0x141, 0xf>(key);  // row_half_mirror
            key = dpp_max_u64<0x140, 0xf>(key);  // row_mirror
            key = dpp_max_u64<0x142, 0xa>(key);  // row_bcast:15 into rows 1 and 3
            key = dpp_max_u64<0x143, 0xc>(key);  // row_bcast:31 into rows 2 and 3
            const int pl = 63 - (__builtin_amdgcn_readlane(static_cast<int>(key), 63) & 63);  // pivot lane
            double prow[K];
#pragma unroll
            for (int j = col; j < K; ++j) {
                const long long bits = __double_as_longlong(arow[j]);
                const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), pl);
                const int hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), pl);
                prow[j] = __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
            }
            if (fabs(prow[col]) <= tiny) {  // singular: numpy raises LinAlgError -> inf (parrm.py:627-628)
                sing = true;
                break;
            }
            if (lane == pl) {
                done = true;
                rank = col;
            } else if (!done) {
                const double l = arow[col] / prow[col];
                arow[col] = l;
#pragma unroll
                for (int j = col + 1; j < K; ++j) arow[j] -= l * prow[j];
            }
        }
        if (sing) {
            if (lane == 0) singular = 1;
        } else if (lane < K) {
#pragma unroll
            for (int j = 0; j < K; ++j) LU[rank * KE + j] = arow[j];
            perm[rank] = lane;
        }
    }
    __syncthreads();
    if (singular) {
        if constexpr (NW == 4) {
            if (tid == 0) err[p] = INFINITY;
        } else {
            if (lane == 0)
                const_cast<double *>(base)[(blk.y >> 2) * elems + KP * kNCol + kYCols + quarter] = INFINITY;
        }
        return;
    }

    const double ksum = static_cast<double>(K * (K + 1) / 2);
    double total = 0.0;
    const int zb_lo = NW == 4 ? 0 : static_cast<int>(blk.y >> 2), zb_hi = NW == 4 ? nz : zb_lo + 1;
    for (int zb = zb_lo; zb < zb_hi; ++zb) {
        const double *rz = base + zb * elems;
        const int ncol = min(kYCols, n_chans - zb * kYCols);
        double e_c = 0.0;
        if (ch < ncol) {
            // b = P r: the permutation is wave-uniform, so each entry is fetched from its permuted
            // row (L2-resident); r itself is re-read for the quadratic form
            double b[K];
#pragma unroll
            for (int i = 0; i < K; ++i) b[i] = rz[perm[i] * kNCol + ch];
            // the factors are wave-uniform: every entry is an LDS broadcast read, so they are
            // fetched as aligned pairs (one ds_read_b128 per two multiply-adds)
            auto pair_at = [&](int i, int j) -> double2 {  // j even
                return *reinterpret_cast<const double2 *>(&LU[i * KE + j]);
            };
#pragma unroll
            for (int i = 1; i < K; ++i) {  // L z = P r
                double v0 = b[i], v1 = 0.0;
#pragma unroll
                for (int j = 0; j + 1 < i; j += 2) {
                    const double2 l = pair_at(i, j);
                    v0 = fma(-l.x, b[j], v0);
                    v1 = fma(-l.y, b[j + 1], v1);
                }
                if (i & 1) v0 = fma(-LU[i * KE + i - 1], b[i - 1], v0);
                b[i] = v0 + v1;
            }
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {  // U beta = z
                double v0 = b[i], v1 = 0.0;
                const int first = (i + 2) & ~1;  // first even column > i
                if (first != i + 1 && i + 1 < K) v0 = fma(-LU[i * KE + i + 1], b[i + 1], v0);
#pragma unroll
                for (int j = first; j + 1 < K; j += 2) {
                    const double2 u = pair_at(i, j);
                    v0 = fma(-u.x, b[j], v0);
                    v1 = fma(-u.y, b[j + 1], v1);
                }
                if (first < K && ((K - first) & 1)) v0 = fma(-LU[i * KE + K - 1], b[K - 1], v0);
                b[i] = (v0 + v1) / LU[i * KE + i];
            }
            // |y - W beta|^2 = y'y - 2 beta'r + beta'G beta, and beta'G beta = beta'r up to the LU's
            // backward error (relative eps * cond(G); G is a Gram matrix of near-orthogonal sinusoids),
            // i.e. to the rounding the three-term form carries anyway -- so the K x K quadratic form,
            // half of this phase, is not evaluated
            double t1 = 0.0, reg = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                t1 = fma(b[i], rz[i * kNCol + ch], t1);
                reg += (lambda * static_cast<double>(i + 1) / ksum) * (b[i] * b[i]);  // :585-586,:595
            }
            const double yy = rz[KP * kNCol + ch];
            e_c = (yy - t1) / static_cast<double>(n_idx) + reg;
        }
        const double v = parrm::wave_sum(e_c);
        if constexpr (NW == 4) {
            __syncthreads();
            if (lane == 0) wsum[tid >> 6] = v;
            __syncthreads();
            total += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        } else {
            if (lane == 0) const_cast<double *>(rz)[KP * kNCol + kYCols + quarter] = v;
        }
    }
    if constexpr (NW == 4) {
        if (tid == 0) err[p] = total / static_cast<double>(n_chans);  // :597
    }
}
template <int K, int NW>
__global__ void __launch_bounds__(64 * NW) fit_solve_fast_kernel(const double *red, int n_idx, int n_chans, int KP,
                                                                   int nz, double lambda, double *err, int special) {
    fit_solve_fast_body<K, NW>(red, n_idx, n_chans, KP, nz, lambda, err, this_block(), special);
}

// the channel mean of the NW = 1 form: quarters added as the NW = 4 form adds its waves, blocks in order; then, for
// the optimiser's batches, the hand-off of fit_publish_kernel (below) in the same launch
__global__ void __launch_bounds__(64) fit_finish_kernel(const double *red, int n, int KP, int nz, int n_chans, double *err,
                                                         double *host_err, volatile unsigned long long *flag,
                                                         unsigned long long seq) {
    const int64_t elems = static_cast<int64_t>(KP + 1) * kNCol;
    for (int p = threadIdx.x; p < n; p += 64) {
        double total = 0.0;
        for (int zb = 0; zb < nz; ++zb) {
            const double *q = red + (static_cast<int64_t>(p) * nz + zb) * elems + KP * kNCol + kYCols;
            total += (q[0] + q[1]) + (q[2] + q[3]);
        }
        const double e = total / static_cast<double>(n_chans);  // :597
        err[p] = e;
        if (host_err) host_err[p] = e;
    }
    if (host_err) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            *flag = seq;
            __threadfence_system();
        }
    }
}

// ---- gang launches: several independent problems per launch (parrm_fit_errors_multi) -----------------------------
// Per-site period estimation (examples/plot_example_dbs_data.py:52-98) advances many small searches in lock-step.
// Launching every search's five kernels on a side stream of its own left the call bound by launch and dispatch
// overhead (~107 us per search and step for single-channel sites).  Here blockIdx.z picks the problem from a
// device table and every block runs the single-problem kernel body on it -- same geometry per problem (the sample
// split is planned per problem, as for a call of its own), same arithmetic, one launch per kernel for the gang.
struct GangProblem {
    const double *y;
    const int64_t *idx;
    const double *periods;
    double *wmat, *part, *red, *err;
    int64_t ldy;
    double lambda;
    int n_idx, n_pad, n_chans, n_periods, nsplit, groups, trig_blocks, err_off;
};

__global__ void __launch_bounds__(256) fit_trig_stacked_gang(const GangProblem *tab, int bw, int kpc, int exact) {
    const GangProblem &q = tab[blockIdx.z];
    if (static_cast<int>(blockIdx.x) >= q.trig_blocks || static_cast<int>(blockIdx.y) >= q.groups) return;
    fit_trig_stacked_body(q.idx, q.n_idx, q.n_pad, q.periods, q.n_periods, bw, kpc, exact, q.wmat,
                          Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), 0, 1});
}

__global__ void __launch_bounds__(384) fit_trig_packed_gang(const GangProblem *tab, int bw, int exact) {
    const GangProblem &q = tab[blockIdx.z];
    if (static_cast<int>(blockIdx.x) >= q.trig_blocks || static_cast<int>(blockIdx.y) >= q.groups) return;
    fit_trig_packed_body(q.idx, q.n_idx, q.n_pad, q.periods, q.n_periods, bw, exact, q.wmat,
                         Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), 0, 1});
}

template <int NW, int CT = 4, int GM = 1>
__global__ void __launch_bounds__(256) fit_accum_mfma_gang(const GangProblem *tab, int kpc, int kreal) {
    const GangProblem &q = tab[blockIdx.z];
    if (static_cast<int>(blockIdx.x) >= q.nsplit || static_cast<int>(blockIdx.y) >= q.groups) return;
    fit_accum_mfma_body<NW, CT, GM>(q.y, q.ldy, q.wmat, q.n_pad, q.n_idx, q.n_chans, q.n_periods, kpc, q.nsplit, q.part,
                                    Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), 0, 1}, kreal);
}

__global__ void __launch_bounds__(256) fit_reduce_gang(const GangProblem *tab, int64_t elems, int ksp) {
    const GangProblem &q = tab[blockIdx.z];
    if (q.nsplit == 1 || static_cast<int>(blockIdx.y) >= q.n_periods) return;
    fit_reduce_body(q.part, q.nsplit, elems, 1, q.n_chans, q.red,
                    Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), 0, 1}, ksp);
}

template <int K, int NW>
__global__ void __launch_bounds__(64 * NW) fit_solve_fast_gang(const GangProblem *tab, int KP, int special) {
    const GangProblem &q = tab[blockIdx.z];
    if (static_cast<int>(blockIdx.x) >= q.n_periods) return;
    fit_solve_fast_body<K, NW>(q.red, q.n_idx, q.n_chans, KP, 1, q.lambda, q.err,
                               Blk{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), 0, 1}, special);
}

// one workgroup for the whole call: quarter sums -> errors where the one-wave solve ran (K = 41), every problem's
// errors into the page-locked hand-off block, then ONE flag
__global__ void __launch_bounds__(64) fit_publish_gang(const GangProblem *tab, int n_problems, const int *quarters_kp,
                                                        double *host_err, volatile unsigned long long *flag,
                                                        unsigned long long seq) {
    for (int pr = 0; pr < n_problems; ++pr) {
        const GangProblem &q = tab[pr];
        const int KP = quarters_kp[pr];  // > 0: the problem's solve left quarter sums (row KP of its blocks)
        for (int p = threadIdx.x; p < q.n_periods; p += 64) {
            double e;
            if (KP > 0) {
                const double *cell = q.red + static_cast<int64_t>(p) * (KP + 1) * kNCol + KP * kNCol + kYCols;
                e = ((cell[0] + cell[1]) + (cell[2] + cell[3])) / static_cast<double>(q.n_chans);  // :597
                q.err[p] = e;
            } else {
                e = q.err[p];
            }
            host_err[q.err_off + p] = e;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *flag = seq;
        __threadfence_system();
    }
}

struct FitGeom {
    int KP, nz, nsplit, n_pad;
    int64_t elems;       // (KP+1)*kNCol
    size_t w_bytes;      // design matrices
    size_t part_bytes;   // all partial Gram blocks
    size_t red_bytes;    // reduced blocks (0 when nsplit == 1: part is used in place)
    bool packed;         // candidates' design rows packed (fit_accum_mfma_body GM = 2), else padded to KP rows each
    int groups;          // 48-row stacks of the design matrix in the planned layout
};

// The packed rows + special-rows Gram form (GM = 2) is taken where the fast solvers exist.  The reference's operation
// order (sincos of the rounded k*a per harmonic) breaks the angle-addition identities at the 1e-10 of its argument
// rounding, so that mode keeps the full product in the padded layout.  PARRM_FIT_FULL_GRAM=1: the padded form always.
static int exact_trig_mode() {
    static const int mode = getenv("PARRM_FIT_EXACT_TRIG") ? atoi(getenv("PARRM_FIT_EXACT_TRIG")) : 0;
    return mode;
}
static bool packed_form(int bw) {
    const int K = 2 * bw + 1;
    return (K == 11 || K == 21 || K == 41) && !exact_trig_mode() && !getenv("PARRM_FIT_FULL_GRAM");
}

// `plan_periods` (>= n_periods): the candidate count the sample split is chosen for.  A slice of a
// larger grid is planned as the whole grid would be, so that every candidate goes through exactly the
// arithmetic (slices per candidate, order of the partial sums) it would see in one call on the whole
// grid -- what makes a candidate-sliced, multi-GPU evaluation bit-identical to the one-GPU one.
FitGeom fit_geometry(int64_t n_idx, int64_t n_chans, int64_t n_periods, int bw, int64_t plan_periods = 0) {
    FitGeom g{};
    g.KP = kp_for(bw);
    g.nz = static_cast<int>((n_chans + kYCols - 1) / kYCols);
    g.n_pad = static_cast<int>((n_idx + kTJ - 1) / kTJ * kTJ);
    const int64_t tiles = g.n_pad / kTJ;
    g.packed = packed_form(bw);
    const int64_t K = 2 * bw + 1;
    auto stacks = [&](int64_t periods, bool packed) {
        return packed ? (periods * K + 47) / 48 : (periods + (48 / g.KP) - 1) / (48 / g.KP);
    };
    const int64_t groups = stacks(n_periods, g.packed);  // workgroups per sample slice
    const int64_t plan_groups = plan_periods > n_periods ? stacks(plan_periods, g.packed) : groups;
    g.groups = static_cast<int>(groups);
    // Sample slices per candidate group.  512 workgroups are resident at once (2 per CU), so the Gram kernel takes
    // ~ceil(workgroups / 512) rounds of (time of a whole-length workgroup / nsplit): e.g. 381 candidates in 326
    // stacks: nsplit 3 -> 978 workgroups = 2 rounds of a third each.  Slicing is not free: every slice leaves a
    // partial block per candidate (33 / 64 / 125 KB at K = 11 / 21 / 41) that the Gram kernel writes and
    // fit_reduce_kernel reads and adds -- ~2.5 transfers of it at ~5 TB/s; at 10 044 candidates two slices cost
    // 0.33 ms, more than the half round they save (measured: 5.28 ms with two slices, 5.05 ms with one).  Both in
    // microseconds: a whole-length workgroup takes ~0.2 us per sample at 12 MFMAs per wave and step (256 columns;
    // the one-wave forms for <= 16 / <= 32 columns a quarter / half of that).  Never below 4 tiles per slice.
    const int64_t max_split = std::max<int64_t>(1, std::min<int64_t>(64, (tiles + 3) / 4));
    const int64_t resident = static_cast<int64_t>(parrm::device_cu_count()) * (n_chans <= 64 ? 4 : 2);  // one-wave workgroups (<= 64 channels): four per CU, else two (512 on a whole MI355X)
    const double t_full = 0.2 * static_cast<double>(g.n_pad) * (n_chans <= 16 ? 0.25 : (n_chans <= 32 ? 0.5 : 1.0));
    const int64_t plan_p = std::max(plan_periods, n_periods);
    const double cols_used = static_cast<double>(std::min<int64_t>(n_chans, kYCols) + 64);
    const double slice_us = static_cast<double>(plan_p) * g.nz * (g.KP + 1) * cols_used * 8.0 * 2.5 / 5.0e6;
    double best = 1e300;
    g.nsplit = 1;
    for (int64_t ns = 1; ns <= max_split; ++ns) {
        // (a last round that fills at most half the slots runs one workgroup per CU, with the matrix pipes to itself:
        // measured ~0.55 of a full round)
        const int64_t wgs = plan_groups * g.nz * ns, tail = wgs % resident;
        const double rounds = static_cast<double>(wgs / resident) + (tail == 0 ? 0.0 : (tail <= resident / 2 ? 0.55 : 1.0));
        // (grids below 2048 candidates keep round 2's rule -- whole rounds, a 0.4 % penalty per slice: their splits
        // differ little either way, and the sum order they imply is the one the 70 reference periods were matched
        // bit for bit with; one of them moves by an ulp under the other rule's splits)
        const double cost = plan_p >= 2048
                                ? rounds / static_cast<double>(ns) * t_full + (ns > 1 ? slice_us * ns : 0.0)
                                : static_cast<double>((wgs + resident - 1) / resident) / static_cast<double>(ns) * (1.0 + 0.004 * ns);
        if (cost < best - 1e-9) {
            best = cost;
            g.nsplit = static_cast<int>(ns);
        }
    }
    if (const char *force = getenv("PARRM_FIT_X_NSPLIT"))  // (timing experiment)
        g.nsplit = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(max_split, atoi(force))));
    g.elems = static_cast<int64_t>(g.KP + 1) * kNCol;
    // whole 48-row stacks (room for the padded layout too: operands the matrix cores cannot take fall back to it)
    g.w_bytes = static_cast<size_t>(std::max(groups, stacks(n_periods, false))) * 48 * g.n_pad * sizeof(double);
    g.part_bytes = static_cast<size_t>(n_periods) * g.nz * g.nsplit * g.elems * sizeof(double);
    g.red_bytes = g.nsplit > 1 ? static_cast<size_t>(n_periods) * g.nz * g.elems * sizeof(double) : 0;
    return g;
}

// ---- the Nelder-Mead refinement as a device-side chain (round 4) ---------------------------------------------------
// The reference's fmin runs (parrm.py:499-517, :545-550) are ~76 dependent batches per search.  Until round 3 every
// batch went Gram -> reduce -> solve -> finish -> host (spin on a flag, take SciPy's decisions, launch the next
// batch): ~15 us of idle device per batch.  Here the refinement's state -- the runs' simplices, the table of known
// values, the batch that is out -- lives in device memory: the kernel that closes a batch (nm_chain_step_kernel, one
// wave) forms the errors, feeds them to parrm_nm_core.h's state machine and leaves the NEXT batch's abscissae and
// launch geometry in NmChainState, where the next batch's kernels read them.  Those kernels are launched with the
// largest grid any batch size needs and blocks beyond the current batch's geometry return at once, so the host
// enqueues batches AHEAD of the device (parrm::nm_chain_run) and only watches a progress word; once the state says
// "finished" every kernel still queued returns at its first instruction.  A batch of P candidates is planned exactly
// as parrm_fit_errors_host plans it (fit_geometry per P, tabulated), runs the same kernel bodies and adds in the same
// order: the errors -- hence SciPy's decisions -- are the host-stepped path's bits, which the host checks by replaying
// the recorded batches through parrm_nm.hip's independent state machine.
constexpr int kChainTable = 1024;      // known values per refinement (typical ~250-400; more: the host-stepped path)
constexpr int kChainHist = 4096;       // evaluations recorded for the host's replay
constexpr int kChainBatches = 1024;    // batches recorded
enum { kChainRunning = 0, kChainFinished = 1, kChainError = 2 };

struct NmChainState {
    parrm_nmcore::Core core;
    int status;              // kChainRunning / kChainFinished / kChainError + code
    int n_periods;           // the batch that is out: candidates, sample slices, 48-row stacks
    int nsplit, groups;
    int batches, hist_used;  // completed so far
    int n_table, arrivals;
    int nsplit_of[parrm_nmcore::kMaxBatch + 1], groups_of[parrm_nmcore::kMaxBatch + 1];  // fit_geometry per batch size
};

struct NmChainHost {  // page-locked, device-mapped: what the host watches and replays
    volatile unsigned long long progress;  // sequence number of the last step kernel that ran
    volatile long long status;             // (run id << 8) | kChain* once the refinement has ended
    double res_x[parrm_nmcore::kMaxRuns], res_f[parrm_nmcore::kMaxRuns];
    int res_its[parrm_nmcore::kMaxRuns], res_calls[parrm_nmcore::kMaxRuns];
    int n_batches, hist_used, error, pad_;
    int batch_sizes[kChainBatches];
    double hist_x[kChainHist], hist_f[kChainHist];
    NmChainState init;                     // staging of the initial state (copied to the device by the stream)
};

// The table of known values as the step kernel sees it: an open-addressing hash table in LDS (2 048 slots for at most
// kChainTable = 1 024 entries), rebuilt at every launch from the append-only arrays in device memory.  A lookup is a
// per-lane operation (1-2 probes), so lane r can run run r's SciPy step (parrm_nmcore::run_advance, the host's code)
// while its neighbours run theirs.
constexpr int kChainSlots = 2048;
constexpr unsigned long long kChainEmpty = ~0ull;  // (a NaN pattern: NaN abscissae are never stored)
struct LdsHash {
    unsigned long long *keys;
    double *vals;
    __device__ static int home(unsigned long long k) {
        k ^= k >> 29;
        k *= 0x9E3779B97F4A7C15ull;
        return static_cast<int>(k >> (64 - 11));
    }
    __device__ int slot_of(double x) const {
        const unsigned long long k = parrm_nmcore::key_of(x);
        int h = home(k);
        for (int probes = 0; probes < kChainSlots; ++probes) {
            const unsigned long long cur = keys[h];
            if (cur == k) return h;
            if (cur == kChainEmpty) return -1;
            h = (h + 1) & (kChainSlots - 1);
        }
        return -1;
    }
    __device__ bool known(double x) const { return slot_of(x) >= 0; }
    __device__ bool find(double x, double *val) const {
        const int h = slot_of(x);
        if (h < 0) return false;
        *val = vals[h];
        return true;
    }
    // (lanes insert different keys at the same time) returns false when the key was there already
    __device__ bool insert_new(double x, double val) {
        const unsigned long long k = parrm_nmcore::key_of(x);
        int h = home(k);
        for (int probes = 0; probes < kChainSlots; ++probes) {
            const unsigned long long old = atomicCAS(&keys[h], kChainEmpty, k);
            if (old == kChainEmpty) {
                vals[h] = val;
                return true;
            }
            if (old == k) {
                vals[h] = val;
                return false;
            }
            h = (h + 1) & (kChainSlots - 1);
        }
        return false;
    }
};

// ascending, one entry per key, of the values the lanes in `valid` hold (parrm_nmcore::sort_unique for a wave):
// out[rank] = value; returns the count.  tmp: 64 doubles of LDS.
__device__ inline int wave_sort_unique(double v, bool valid, double *tmp, double *out) {
    const int lane = threadIdx.x & 63;
    tmp[lane] = v;
    __syncthreads();
    unsigned long long mask = __ballot(valid);
    const unsigned long long kv = parrm_nmcore::key_of(v);
    bool dup = false;
    int rank = 0;
    for (unsigned long long m = mask; m; m &= m - 1) {
        const int j = __ffsll(static_cast<long long>(m)) - 1;
        const double o = tmp[j];
        if (j < lane && parrm_nmcore::key_of(o) == kv) dup = true;
    }
    const bool keep = valid && !dup;
    const unsigned long long kept = __ballot(keep);
    for (unsigned long long m = kept; m; m &= m - 1) {
        const int j = __ffsll(static_cast<long long>(m)) - 1;
        if (tmp[j] < v) ++rank;
    }
    __syncthreads();
    if (keep) out[rank] = v;
    __syncthreads();
    return __popcll(kept);
}

// Closes a batch and opens the next (one wave).  `first`: nothing is out yet -- the launch that takes the initial state
// from the host's block and forms batch 0.
// Errors of the batch that is out: K = 41 (`quarters`): the one-wave solves left quarter sums in row KP of the
// candidates' blocks -- added as fit_finish_kernel adds them; else fit_solve_fast_kernel<K, 4> left err[p].
// The state machine is parrm_nm_core.h's: run_advance / run_wanted / run_lookahead per lane (lane r = run r), the
// generator's bookkeeping (core_produce) restated over ballots.
struct ChainStepLds {
    parrm_nmcore::Core c;
    unsigned long long keys[kChainSlots];
    double vals[kChainSlots];
    double tmp[64], list[64], need[64];
    int fail;
};
__device__ inline void nm_chain_step(ChainStepLds &L, NmChainState *st, unsigned long long *keys_g, double *vals_g,
                                     const double *part, const double *red, const double *err, int KP, int nz, int n_chans,
                                     int quarters, NmChainHost *host, unsigned long long seq, long long run_id, int first,
                                     int p_cap) {
    using namespace parrm_nmcore;
    const int lane = threadIdx.x & 63;
    Core &c = L.c;
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(first ? &host->init.core : &st->core);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&c);
        for (int i = lane; i < static_cast<int>(sizeof(Core) / 8); i += 64) dst[i] = src[i];
        for (int i = lane; i < kChainSlots; i += 64) L.keys[i] = kChainEmpty;
        if (lane == 0) L.fail = 0;
    }
    __syncthreads();
    LdsHash table{L.keys, L.vals};
    int n_table = first ? 0 : st->n_table;
    int batches = first ? 0 : st->batches, used = first ? 0 : st->hist_used;
    const int nsplit_out = first ? 1 : st->nsplit;
    // the errors of the batch that is out are requested before its size is known (lanes beyond it read cells nobody
    // uses: the blocks of p_cap candidates are inside the workspace), so that their latency overlaps the table's
    double e_mine = 0.0;
    if (!first && lane < p_cap) {
        if (quarters) {
            const double *blocks = nsplit_out > 1 ? red : part;
            const int64_t elems = static_cast<int64_t>(KP + 1) * kNCol;
            double total = 0.0;
            for (int zb = 0; zb < nz; ++zb) {
                const double *q = blocks + (static_cast<int64_t>(lane) * nz + zb) * elems + KP * kNCol + kYCols;
                total += (q[0] + q[1]) + (q[2] + q[3]);
            }
            e_mine = total / static_cast<double>(n_chans);  // :597
        } else {
            e_mine = err[lane];
        }
    }
    for (int i = lane; i < n_table; i += 64) table.insert_new(__longlong_as_double(static_cast<long long>(keys_g[i])), vals_g[i]);
    __syncthreads();
    if (!first) {
        const int P = c.n_batch;
        if (batches >= kChainBatches || used + P > kChainHist || n_table + P > kChainTable || P > p_cap) {
            if (lane == 0) L.fail = 3;  // history or table full: the host takes the stepped path
        } else if (lane < P) {
            const double e = e_mine;
            const double x = c.batch[lane];
            host->hist_x[used + lane] = x;
            host->hist_f[used + lane] = e;
            if (lane == 0) host->batch_sizes[batches] = P;
            if (!table.insert_new(x, e)) L.fail = 4;  // (a batch never holds a known abscissa)
            keys_g[n_table + lane] = key_of(x);
            vals_g[n_table + lane] = e;
        }
        __syncthreads();
        if (!L.fail) {
            n_table += P;
            used += P;
            ++batches;
            if (lane == 0) c.state = kAdvance;
        }
        __syncthreads();
    }
    // the generator (parrm_nmcore::core_produce) up to its next batch or its end
    int n_batch = 0;
    while (!L.fail) {
        if (c.state == kDone) break;
        if (c.state == kTop) {
            const bool is_active = lane < c.n_runs && !c.runs[lane].done;
            const unsigned long long active = __ballot(is_active);
            if (active == 0) {
                __syncthreads();
                if (lane == 0) c.state = kDone;
                __syncthreads();
                break;
            }
            double w[3] = {0, 0, 0};
            int nw = 0;
            if (is_active) nw = run_wanted(c.runs[lane], w);
            __syncthreads();
            // the runs' abscissae, ascending, one of each; then those not known yet (order kept)
            int n_want = 0;
            {
                // three rounds (k-th abscissa of every run), merged through the list: values sit in lanes 0 .. 23
                if (lane < 24) L.need[lane] = 0.0;
                __syncthreads();
                if (is_active)
                    for (int k = 0; k < nw; ++k) L.need[3 * lane + k] = w[k];
                __syncthreads();
                bool valid = false;
                double v = 0.0;
                if (lane < 3 * kMaxRuns) {
                    const int r = lane / 3, k = lane - 3 * r;
                    const bool act = (active >> r) & 1ull;
                    const int cnt = act ? (c.runs[r].iterations == 0 ? 2 : 3) : 0;
                    v = L.need[lane];
                    valid = k < cnt && !is_nan(v);
                }
                __syncthreads();
                n_want = wave_sort_unique(v, valid, L.tmp, L.list);
            }
            const bool asks = lane < n_want;
            const double xq = asks ? L.list[lane] : 0.0;
            const bool unknown = asks && !table.known(xq);
            const unsigned long long unk = __ballot(unknown);
            int n_need = __popcll(unk);
            __syncthreads();
            if (unknown) L.need[__popcll(unk & ((1ull << lane) - 1ull))] = xq;
            if (lane == 0) {
                c.pending = static_cast<unsigned>(active);
                c.state = kAdvance;
            }
            __syncthreads();
            if (n_need > 0) {
                if (__popcll(active) <= c.lookahead_runs) {
                    // the likely following step's abscissae of every run in flight: lane 8 r + k holds the k-th of run r
                    const int r = lane >> 3, k = lane & 7;
                    bool valid = false;
                    double v = 0.0;
                    if (r < c.n_runs && ((active >> r) & 1ull) && k < 6) {
                        double a[6];
                        const int n = run_lookahead(c.runs[r], a);
                        if (k < n) {
                            v = a[k];
                            valid = !is_nan(v) && !table.known(v);
                        }
                    }
                    // appended behind the wanted ones, then everything sorted again (as the host does)
                    const unsigned long long extra = __ballot(valid);
                    __syncthreads();
                    if (valid) L.list[n_need + __popcll(extra & ((1ull << lane) - 1ull))] = v;
                    if (lane < n_need) L.list[lane] = L.need[lane];
                    __syncthreads();
                    const int total = n_need + __popcll(extra);
                    const bool has = lane < total;
                    const double vv = has ? L.list[lane] : 0.0;
                    __syncthreads();
                    n_need = wave_sort_unique(vv, has, L.tmp, L.need);
                }
                if (n_need > p_cap) {  // (the batch kernels' grids are cut for p_cap candidates)
                    if (lane == 0) L.fail = 4 + kErrBatchTooLarge;
                    __syncthreads();
                    break;
                }
                if (lane < n_need) c.batch[lane] = L.need[lane];
                if (lane == 0) {
                    c.n_batch = n_need;
                    c.state = kWait;
                }
                n_batch = n_need;
                __syncthreads();
                break;
            }
        }
        // kAdvance: the steps of the pending runs, lane r = run r
        const bool due = lane < c.n_runs && ((c.pending >> lane) & 1u);
        double m = 0.0;
        bool missed = false;
        if (due) missed = run_advance(c.runs[lane], table, &m) != 0;
        const unsigned long long still = __ballot(missed);
        __syncthreads();
        if (still == 0) {
            if (lane == 0) c.state = kTop;
            __syncthreads();
            continue;
        }
        // expansion / shrink points of the few runs that need them
        const int n_missing = wave_sort_unique(m, missed, L.tmp, L.need);
        if (lane < n_missing) c.batch[lane] = L.need[lane];
        if (lane == 0) {
            c.n_batch = n_missing;
            c.pending = static_cast<unsigned>(still);
            c.state = kWait;
        }
        n_batch = n_missing;
        __syncthreads();
        break;
    }
    __syncthreads();
    const int fail = L.fail;
    const bool ended = fail || c.state == kDone;
    if (ended && lane == 0) c.n_batch = 0;
    __syncthreads();
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&c);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&st->core);
        for (int i = lane; i < static_cast<int>(sizeof(Core) / 8); i += 64) dst[i] = src[i];
    }
    if (lane == 0) {
        const int P = ended ? 0 : n_batch;
        const int *nsplit_of = first ? host->init.nsplit_of : st->nsplit_of;
        const int *groups_of = first ? host->init.groups_of : st->groups_of;
        st->n_periods = P;
        st->nsplit = nsplit_of[P];
        st->groups = groups_of[P];
        st->n_table = n_table;
        st->batches = batches;
        st->hist_used = used;
        st->arrivals = 0;
        st->status = ended ? (fail ? kChainError + fail : kChainFinished) : kChainRunning;
    }
    if (first)
        for (int i = lane; i <= kMaxBatch; i += 64) {
            st->nsplit_of[i] = host->init.nsplit_of[i];
            st->groups_of[i] = host->init.groups_of[i];
        }
    if (ended) {
        if (lane < c.n_runs) {
            double x, f;
            int its, calls;
            run_result(c.runs[lane], &x, &f, &its, &calls);
            host->res_x[lane] = x;
            host->res_f[lane] = f;
            host->res_its[lane] = its;
            host->res_calls[lane] = calls;
        }
        if (lane == 0) {
            host->n_batches = batches;
            host->hist_used = used;
            host->error = fail;
        }
    }
    // The host reads the history and the results only once the status word says the refinement has ended: that is the
    // one place where the order of the stores to its block matters.  In between, `progress` only paces its queue.
    if (ended) {
        __threadfence_system();
        __syncthreads();
        if (lane == 0) {
            host->status = (run_id << 8) | (fail ? kChainError : kChainFinished);
            __threadfence_system();
        }
    }
    if (lane == 0) host->progress = seq;
}
__global__ void __launch_bounds__(64) nm_chain_step_kernel(NmChainState *st, unsigned long long *keys_g, double *vals_g,
                                                            const double *part, const double *red, const double *err, int KP,
                                                            int nz, int n_chans, int quarters, NmChainHost *host,
                                                            unsigned long long seq, long long run_id, int first, int p_cap) {
    if (!first && st->status != kChainRunning) return;
    __shared__ ChainStepLds L;
    nm_chain_step(L, st, keys_g, vals_g, part, red, err, KP, nz, n_chans, quarters, host, seq, run_id, first, p_cap);
}

// The batch's kernels: the bodies of the plain kernels on the geometry the state names.
// The batch's abscissae are read through the CONSTANT address space, as the plain kernels read theirs from the kernarg
// segment: the state does not change while a batch kernel runs, and as an ordinary global pointer the compiler must
// assume the kernel's own stores alias it -- it then fetches the periods with vector loads, which sit in the Gram
// kernel's hand-counted vmcnt queue (measured: +20 us per Gram launch of an optimiser batch).
typedef const double __attribute__((address_space(4))) * ConstDoublePtr;
__device__ __forceinline__ const double *chain_periods(const NmChainState *st) {
    return (const double *)(ConstDoublePtr)(st->core.batch);  // (C casts: they cross address spaces)
}
// A Gram launch numbers its (sample slice, stack) blocks in ONE dimension and cuts that number by the batch's own slice
// count: the blocks that have work are then the FIRST nsplit x groups of the launch, in the order a launch cut for the
// batch would dispatch them, and the surplus comes last.  (With a two-dimensional grid of the widest geometry the
// surplus blocks of every row sat between the working ones: they return at once, but the slots they free are refilled
// out of step, some CUs end up with three or four working blocks and others with one -- measured +45 ... +130 us on
// the batches whose slice count is below the widest, 56 / 51 / 46 / 39 against 64.)
__device__ __forceinline__ Blk chain_gram_block(int nsplit) {
    const int id = static_cast<int>(blockIdx.x);
    const int y = id / nsplit;
    return Blk{id - y * nsplit, y, static_cast<int>(blockIdx.z), static_cast<int>(gridDim.z)};
}
__global__ void __launch_bounds__(384) nm_chain_trig_kernel(const NmChainState *st, const int64_t *idx, int n_idx, int n_pad,
                                                             int bw, double *Ws) {
    if (st->status != kChainRunning || static_cast<int>(blockIdx.y) >= st->groups) return;
    fit_trig_packed_body(idx, n_idx, n_pad, chain_periods(st), st->n_periods, bw, 0, Ws, this_block());
}
__global__ void __launch_bounds__(256) nm_chain_fused_kernel(const NmChainState *st, const double *Y, int64_t ldy,
                                                              const int64_t *idx, int n_pad, int n_idx, int n_chans, int kpc,
                                                              double *part, int kreal) {
    if (st->status != kChainRunning) return;
    const int nsplit = st->nsplit;
    const Blk blk = chain_gram_block(nsplit);
    if (blk.y >= st->groups) return;
    fit_accum_fused_body(Y, ldy, idx, chain_periods(st), n_pad, n_idx, n_chans, st->n_periods, kpc, nsplit, part, blk, kreal);
}
template <int NW, int CT>
__global__ void __launch_bounds__(256) nm_chain_accum_kernel(const NmChainState *st, const double *Y, int64_t ldy,
                                                              const double *Ws, int n_pad, int n_idx, int n_chans, int kpc,
                                                              double *part, int kreal) {
    if (st->status != kChainRunning) return;
    const int nsplit = st->nsplit;
    const Blk blk = chain_gram_block(nsplit);
    if (blk.y >= st->groups) return;
    fit_accum_mfma_body<NW, CT, 2>(Y, ldy, Ws, n_pad, n_idx, n_chans, st->n_periods, kpc, nsplit, part, blk, kreal);
}
__global__ void __launch_bounds__(256) nm_chain_reduce_kernel(const NmChainState *st, const double *part, int64_t elems, int nz,
                                                               int n_chans, double *red, int ksp) {
    if (st->status != kChainRunning) return;
    const int nsplit = st->nsplit;
    if (nsplit <= 1 || static_cast<int>(blockIdx.y) >= st->n_periods * nz) return;
    fit_reduce_body(part, nsplit, elems, nz, n_chans, red, this_block(), ksp);
}
template <int K, int NW>
__global__ void __launch_bounds__(64 * NW) nm_chain_solve_kernel(const NmChainState *st, const double *part, const double *red,
                                                                   int n_idx, int n_chans, int KP, int nz, double lambda,
                                                                   double *err) {
    if (st->status != kChainRunning || static_cast<int>(blockIdx.x) >= st->n_periods) return;
    fit_solve_fast_body<K, NW>(st->nsplit > 1 ? red : part, n_idx, n_chans, KP, nz, lambda, err, this_block(), 1);
}

// ---- host hand-off of the small optimiser batches ---------------------------------------------
// One Nelder-Mead step is a handful of candidates; the D2H copy + stream synchronise of the plain
// path cost more host time (~40 us) than some of the kernels.  Instead a one-wave kernel at the end
// of the stream copies the errors into a pinned, device-mapped staging block and then raises a
// sequence flag (system-scope release) that the host spins on.
__global__ void __launch_bounds__(64) fit_publish_kernel(const double *err, int n, double *host_err,
                                                          volatile unsigned long long *flag,
                                                          unsigned long long seq) {
    for (int i = threadIdx.x; i < n; i += 64) host_err[i] = err[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *flag = seq;
        __threadfence_system();
    }
}

struct HostStage {
    static constexpr int64_t kCap = 4096;  // candidates per staged call; larger batches use the copy path
    static constexpr int kMaxProblems = 64;  // problems per parrm_fit_errors_multi call
    static constexpr int kStreams = 8;       // side streams the problems of one call are dealt over
    double *h = nullptr;                   // [kCap] periods (multi) | [kCap] errors | flag | [kMaxProblems] flags (multi)
    double *d = nullptr;                   // the same block through the device's mapping
    // gang launches: problem table | quarter flags | periods, page-locked on the host and its device copy
    static constexpr size_t kGangBytes = kMaxProblems * (sizeof(GangProblem) + sizeof(int)) + kCap * sizeof(double);
    unsigned char *h_gang = nullptr, *d_gang = nullptr;
    // device-side Nelder-Mead chain: progress word, results and the history the host replays (mapped, see NmChainHost)
    NmChainHost *h_chain = nullptr, *d_chain = nullptr;
    long long chain_runs = 0;
    int device = -1;
    unsigned long long seq = 0;
    hipStream_t side[kStreams] = {};
    hipEvent_t fork = nullptr, join[kStreams] = {};
    void release() {
        for (int i = 0; i < kStreams; ++i) {
            if (join[i]) (void)hipEventDestroy(join[i]);
            if (side[i]) (void)hipStreamDestroy(side[i]);
            join[i] = nullptr;
            side[i] = nullptr;
        }
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
        if (h) (void)hipHostFree(h);
        h = d = nullptr;
        if (h_gang) (void)hipHostFree(h_gang);
        if (d_gang) (void)hipFree(d_gang);
        h_gang = d_gang = nullptr;
        if (h_chain) (void)hipHostFree(h_chain);
        h_chain = d_chain = nullptr;
        device = -1;
    }
    bool chain_ready() {
        if (h_chain) return true;
        void *p = nullptr, *dp = nullptr;
        if (hipHostMalloc(&p, sizeof(NmChainHost), hipHostMallocMapped) != hipSuccess) return false;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
            (void)hipHostFree(p);
            return false;
        }
        std::memset(p, 0, sizeof(NmChainHost));
        h_chain = static_cast<NmChainHost *>(p);
        d_chain = static_cast<NmChainHost *>(dp);
        return true;
    }
    bool gang_ready() {
        if (h_gang && d_gang) return true;
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, kGangBytes, hipHostMallocDefault) != hipSuccess) return false;
        if (hipMalloc(&dp, kGangBytes) != hipSuccess) {
            (void)hipHostFree(hp);
            return false;
        }
        h_gang = static_cast<unsigned char *>(hp);
        d_gang = static_cast<unsigned char *>(dp);
        return true;
    }
    bool streams_ready() {
        if (fork) return true;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        for (int i = 0; i < kStreams; ++i) {
            if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        return true;
    }
    bool ready(int dev) {
        if (h && device == dev) return true;
        release();
        void *p = nullptr;
        if (hipHostMalloc(&p, (2 * kCap + 8 + kMaxProblems) * sizeof(double), hipHostMallocMapped) != hipSuccess) return false;
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
            (void)hipHostFree(p);
            return false;
        }
        h = static_cast<double *>(p);
        d = static_cast<double *>(dp);
        device = dev;
        for (int i = 0; i < 8 + kMaxProblems; ++i) *reinterpret_cast<volatile unsigned long long *>(h + 2 * kCap + i) = 0;
        return true;
    }
};

// Ownership of the hand-off blocks.  A block lives as long as the thread that uses it may call in
// again, i.e. for the process: it is owned by this registry (one heap object per calling thread, found
// through a thread_local POINTER -- a trivially destructible thread_local, so nothing of ours runs in
// the thread-exit / process-exit destructor phase) and released by parrm_hip_shutdown(), which the
// host binding calls while the HIP runtime is certainly still up (Python: an atexit hook, i.e. before
// interpreter and library teardown).  Round 1 first freed the block from a thread_local destructor and
// then leaked it; see DESIGN.md "Teardown".
struct StageRegistry {
    std::mutex mu;
    std::vector<HostStage *> all;
};
StageRegistry &registry() {
    static StageRegistry *r = new StageRegistry();  // never destroyed: no static-destructor ordering to get wrong
    return *r;
}
thread_local HostStage *t_stage = nullptr;
HostStage &thread_stage() {
    if (!t_stage) {
        t_stage = new HostStage();
        StageRegistry &r = registry();
        std::lock_guard<std::mutex> lock(r.mu);
        r.all.push_back(t_stage);
    }
    return *t_stage;
}

}  // namespace

extern "C" {

size_t parrm_absdiff_workspace_bytes(int64_t n_chans, int64_t n_samples) {
    if (n_chans <= 0 || n_samples <= 1) return sizeof(double);
    const int64_t n_chunk = (n_samples - 1 + kStatChunk - 1) / kStatChunk;
    return static_cast<size_t>(n_chans * n_chunk) * sizeof(double);
}

int parrm_absdiff_mean(const void *d_x, int x_dtype, int64_t n_chans, int64_t n_samples, int64_t ldx,
                       double *d_scale, void *d_workspace, size_t workspace_bytes, void *stream) {
    PARRM_REQUIRE(d_x && d_scale && d_workspace, "absdiff_mean: NULL argument");
    PARRM_REQUIRE(x_dtype == PARRM_F32 || x_dtype == PARRM_F64, "absdiff_mean: bad dtype %d", x_dtype);
    PARRM_REQUIRE(n_chans > 0 && n_samples >= 2 && ldx >= n_samples, "absdiff_mean: bad shape");
    if (workspace_bytes < parrm_absdiff_workspace_bytes(n_chans, n_samples)) {
        parrm::set_error("absdiff_mean: workspace too small");
        return PARRM_ERR_WORKSPACE;
    }
    const int64_t n_chunk = (n_samples - 1 + kStatChunk - 1) / kStatChunk;
    const int64_t blocks = n_chans * n_chunk;
    PARRM_REQUIRE(blocks <= 0x7fffffffLL, "absdiff_mean: recording too large for one launch");
    double *partial = static_cast<double *>(d_workspace);
    hipStream_t s = parrm::as_stream(stream);
    if (x_dtype == PARRM_F64)
        hipLaunchKernelGGL(absdiff_partial_kernel<double>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                           static_cast<const double *>(d_x), n_samples, ldx, n_chunk, partial);
    else
        hipLaunchKernelGGL(absdiff_partial_kernel<float>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                           static_cast<const float *>(d_x), n_samples, ldx, n_chunk, partial);
    PARRM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(absdiff_final_kernel, dim3(static_cast<unsigned>(n_chans)), dim3(64), 0, s, partial, n_chunk,
                       n_samples, d_scale);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

int parrm_gather_standardise(const void *d_x, int x_dtype, int64_t n_chans, int64_t n_samples, int64_t ldx,
                             const int64_t *d_idx, int64_t n_idx, const double *d_scale,
                             double outlier_boundary, double *d_y, int64_t ldy, void *stream) {
    PARRM_REQUIRE(d_x && d_idx && d_scale && d_y, "gather_standardise: NULL argument");
    PARRM_REQUIRE(x_dtype == PARRM_F32 || x_dtype == PARRM_F64, "gather_standardise: bad dtype %d", x_dtype);
    PARRM_REQUIRE(n_chans > 0 && n_samples >= 2 && n_idx > 0 && ldy >= n_chans && ldx >= n_samples,
                  "gather_standardise: bad shape");
    const dim3 grid(static_cast<unsigned>((n_idx + 255) / 256), static_cast<unsigned>(std::min<int64_t>(n_chans, 65535)));
    hipStream_t s = parrm::as_stream(stream);
    if (x_dtype == PARRM_F64)
        hipLaunchKernelGGL(gather_kernel<double>, grid, dim3(256), 0, s, static_cast<const double *>(d_x), n_chans,
                           ldx, d_idx, n_idx, d_scale, outlier_boundary, d_y, ldy);
    else
        hipLaunchKernelGGL(gather_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(d_x), n_chans,
                           ldx, d_idx, n_idx, d_scale, outlier_boundary, d_y, ldy);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

size_t parrm_fit_workspace_bytes(int64_t n_idx, int64_t n_chans, int64_t n_periods, int bw) {
    return parrm_fit_slice_workspace_bytes(n_idx, n_chans, n_periods, n_periods, bw);
}

size_t parrm_fit_slice_workspace_bytes(int64_t n_idx, int64_t n_chans, int64_t n_periods, int64_t grid_periods,
                                       int bw) {
    if (n_idx <= 0 || n_chans <= 0 || n_periods <= 0 || grid_periods < n_periods || bw < 0 || bw > kMaxBw) return 0;
    const FitGeom g = fit_geometry(n_idx, n_chans, n_periods, bw, grid_periods);
    return g.w_bytes + g.part_bytes + g.red_bytes;
}

int parrm_fit_errors(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx, int64_t n_chans,
                     const double *d_periods, int64_t n_periods, int bw, double lambda, double *d_err,
                     void *d_workspace, size_t workspace_bytes, void *stream) {
    return parrm_fit_errors_slice(d_y, ldy, d_idx, n_idx, n_chans, d_periods, n_periods, n_periods, bw, lambda, d_err,
                                  d_workspace, workspace_bytes, stream);
}

// hand-off of an optimiser batch's errors to the host (parrm_fit_errors_host): when the batch takes the one-wave
// solve, its finishing kernel publishes too and `done` is set

struct Publish {
    double *host_err;
    volatile unsigned long long *flag;
    unsigned long long seq;
    bool done;
    const double *h_periods;  // the batch's periods on the host (<= kSmallPeriods: passed by value, d_periods unused)
};
static int fit_errors_impl(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx, int64_t n_chans,
                           const double *d_periods, int64_t n_periods, int64_t grid_periods, int bw, double lambda,
                           double *d_err, void *d_workspace, size_t workspace_bytes, void *stream, Publish *pub);

int parrm_fit_errors_slice(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx, int64_t n_chans,
                           const double *d_periods, int64_t n_periods, int64_t grid_periods, int bw, double lambda,
                           double *d_err, void *d_workspace, size_t workspace_bytes, void *stream) {
    return fit_errors_impl(d_y, ldy, d_idx, n_idx, n_chans, d_periods, n_periods, grid_periods, bw, lambda, d_err,
                           d_workspace, workspace_bytes, stream, nullptr);
}

static int fit_errors_impl(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx, int64_t n_chans,
                           const double *d_periods, int64_t n_periods, int64_t grid_periods, int bw, double lambda,
                           double *d_err, void *d_workspace, size_t workspace_bytes, void *stream, Publish *pub) {
    PARRM_REQUIRE(d_y && d_idx && d_periods && d_err && d_workspace, "fit_errors: NULL argument");
    PARRM_REQUIRE(grid_periods >= n_periods, "fit_errors: a slice cannot be longer than its grid");
    PARRM_REQUIRE(bw >= 0 && bw <= kMaxBw, "fit_errors: bandwidth %d outside [0, %d]", bw, kMaxBw);
    PARRM_REQUIRE(n_idx > 0 && n_idx < (int64_t{1} << 30) && n_chans > 0 && n_chans < (int64_t{1} << 24) &&
                      ldy >= n_chans,
                  "fit_errors: bad shape");
    PARRM_REQUIRE(n_periods > 0 && n_periods <= 65535, "fit_errors: 1..65535 periods per call");
    const FitGeom g = fit_geometry(n_idx, n_chans, n_periods, bw, grid_periods);
    if (workspace_bytes < g.w_bytes + g.part_bytes + g.red_bytes) {
        parrm::set_error("fit_errors: workspace too small (%zu < %zu)", workspace_bytes,
                         g.w_bytes + g.part_bytes + g.red_bytes);
        return PARRM_ERR_WORKSPACE;
    }
    PARRM_REQUIRE(g.nz <= 65535, "fit_errors: too many channels");
    hipStream_t s = parrm::as_stream(stream);
    double *wmat = static_cast<double *>(d_workspace);
    double *part = wmat + g.w_bytes / sizeof(double);
    double *red = g.nsplit > 1 ? part + g.part_bytes / sizeof(double) : part;
    const int n = static_cast<int>(n_idx), C = static_cast<int>(n_chans), P = static_cast<int>(n_periods);
    const int exact_trig = exact_trig_mode();
    const int ncb = 48 / g.KP;
    // matrix-core path: needs 16-byte aligned column quads (even ldy) and 31-bit byte offsets; else the
    // vector-ALU kernel
    const char *accum_env = getenv("PARRM_FIT_ACCUM");  // 1 = vector-ALU kernel (read per call: tests A/B the two)
    const int accum_choice = accum_env ? atoi(accum_env) : 0;
    const bool mfma_ok = ldy % 2 == 0 && (reinterpret_cast<uintptr_t>(d_y) & 15) == 0 &&
                         n_idx * ldy * 8 < 0x7fff0000LL && static_cast<int64_t>(g.n_pad) * 48 * 8 * 3 < 0x7fff0000LL;
    const bool use_mfma = accum_choice == 1 ? false : mfma_ok;
    const int K = 2 * bw + 1;
    const bool special = use_mfma && g.packed;  // packed rows, W'W from three of its rows (fit_accum_mfma_body GM = 2)
    const int groups = special ? g.groups : (P + ncb - 1) / ncb;
    SmallPeriods small{};
    const int by_value = pub && pub->h_periods && P <= kSmallPeriods;
    if (by_value) std::memcpy(small.p, pub->h_periods, static_cast<size_t>(P) * sizeof(double));
    // 256-channel blocks of the packed form: the Gram kernel computes the design rows itself (fit_accum_fused_body);
    // PARRM_FIT_UNFUSED=1 keeps the design-matrix kernel + fit_accum_mfma_kernel<4, 4, 2> (same bits)
    const bool wide = use_mfma && C > 64 && !(C <= 128 && g.nz == 1 && !getenv("PARRM_FIT_NO_TWO_WAVES"));
    const bool fused = special && wide && !getenv("PARRM_FIT_UNFUSED");
    if (fused) {
        // (the attribute belongs to the function ON A DEVICE: once per device of a multi-device process)
        static std::atomic<bool> lds_allowed[64];
        int dev = 0;
        PARRM_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !lds_allowed[dev].load(std::memory_order_acquire)) {
            PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fit_accum_fused_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLdsBytes));
            if (dev >= 0 && dev < 64) lds_allowed[dev].store(true, std::memory_order_release);
        }
    } else if (special) {
        hipLaunchKernelGGL(fit_trig_packed_kernel, dim3((g.n_pad + 63) / 64, groups), dim3(64 * packed_trig_waves(K)), 0, s,
                           small, d_idx, n, g.n_pad, d_periods, P, bw, exact_trig, wmat, by_value);
    } else if (use_mfma) {
        hipLaunchKernelGGL(fit_trig_stacked_kernel, dim3((g.n_pad + 63) / 64, groups), dim3(64 * ncb), 0, s, small, d_idx, n,
                           g.n_pad, d_periods, P, bw, g.KP, exact_trig, wmat, by_value);
    } else {
        hipLaunchKernelGGL(fit_trig_kernel, dim3((g.n_pad + 255) / 256, P), dim3(256), 0, s, small, d_idx, n, g.n_pad,
                           d_periods, bw, g.KP, exact_trig, wmat, by_value);
    }
    PARRM_HIP_CHECK(hipGetLastError());
    const dim3 grid(g.nsplit, groups, g.nz);
    const bool narrow_ok = !getenv("PARRM_FIT_NO_NARROW16");  // (A/B knob: the 64-column form for every C <= 64)
#define PARRM_LAUNCH_ACCUM(NW_, CT_, THREADS_)                                                                              \
    do {                                                                                                                   \
        if (special)                                                                                                       \
            hipLaunchKernelGGL((fit_accum_mfma_kernel<NW_, CT_, 2>), grid, dim3(THREADS_), 0, s, d_y, ldy, wmat, g.n_pad, n, C, \
                               P, g.KP, g.nsplit, part, K);                                                                \
        else                                                                                                               \
            hipLaunchKernelGGL((fit_accum_mfma_kernel<NW_, CT_, 1>), grid, dim3(THREADS_), 0, s, d_y, ldy, wmat, g.n_pad, n, C, \
                               P, g.KP, g.nsplit, part, K);                                                                \
    } while (0)
    if (fused)
        hipLaunchKernelGGL(fit_accum_fused_kernel, grid, dim3(256), kFusedLdsBytes, s, small, d_y, ldy, d_idx, d_periods, g.n_pad,
                           n, C, P, g.KP, g.nsplit, part, K, by_value);
    else if (use_mfma && C <= 16 && narrow_ok)
        PARRM_LAUNCH_ACCUM(1, 1, 64);
    else if (use_mfma && C <= 32 && narrow_ok)
        PARRM_LAUNCH_ACCUM(1, 2, 64);
    else if (use_mfma && C <= 64)
        PARRM_LAUNCH_ACCUM(1, 4, 64);
    else if (use_mfma && C <= 128 && g.nz == 1 && !getenv("PARRM_FIT_NO_TWO_WAVES"))
        PARRM_LAUNCH_ACCUM(2, 4, 128);
    else if (use_mfma)
        PARRM_LAUNCH_ACCUM(4, 4, 256);
    else
        hipLaunchKernelGGL(fit_accum_kernel, grid, dim3(256), 0, s, d_y, ldy, wmat, g.n_pad, n, C, P, g.KP, g.nsplit, part);
#undef PARRM_LAUNCH_ACCUM
    PARRM_HIP_CHECK(hipGetLastError());
    if (g.nsplit > 1) {
        const dim3 rgrid(static_cast<unsigned>((g.elems + 255) / 256), static_cast<unsigned>(n_periods * g.nz));
        PARRM_REQUIRE(n_periods * g.nz <= 65535, "fit_errors: periods x column blocks too large");
        hipLaunchKernelGGL(fit_reduce_kernel, rgrid, dim3(256), 0, s, part, g.nsplit, g.elems, g.nz, C, red, special ? K : 0);
        PARRM_HIP_CHECK(hipGetLastError());
    }
    const int sp = special ? 1 : 0;
    if (K == 11 || K == 21 || K == 41) {
        // small batches (the optimiser's steps): one wave per 64-channel quarter, see fit_solve_fast_kernel
        // (measured per batch, 256 channels: K = 41 222 -> 208 us at 9 candidates, 155 -> 141 us at 4; K = 21 and 11
        // unchanged or 3 us worse -- their factorisation is short and the extra launch shows -- so K = 41 only)
        const bool quarters = K == 41 && n_periods <= 64 && 4 * g.nz <= 65535 && !getenv("PARRM_FIT_SOLVE_ONE_WORKGROUP");
        if (quarters) {
            const dim3 sg(static_cast<unsigned>(n_periods), static_cast<unsigned>(4 * g.nz)), sb(64);
            if (K == 11)
                hipLaunchKernelGGL((fit_solve_fast_kernel<11, 1>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
            else if (K == 21)
                hipLaunchKernelGGL((fit_solve_fast_kernel<21, 1>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
            else
                hipLaunchKernelGGL((fit_solve_fast_kernel<41, 1>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
            PARRM_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(fit_finish_kernel, dim3(1), dim3(64), 0, s, red, P, g.KP, g.nz, C, d_err,
                               pub ? pub->host_err : nullptr, pub ? pub->flag : nullptr, pub ? pub->seq : 0ull);
            PARRM_HIP_CHECK(hipGetLastError());
            if (pub) pub->done = true;
            return PARRM_OK;
        }
        const dim3 sg(static_cast<unsigned>(n_periods)), sb(256);
        if (K == 11)
            hipLaunchKernelGGL((fit_solve_fast_kernel<11, 4>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
        else if (K == 21)
            hipLaunchKernelGGL((fit_solve_fast_kernel<21, 4>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
        else
            hipLaunchKernelGGL((fit_solve_fast_kernel<41, 4>), sg, sb, 0, s, red, n, C, g.KP, g.nz, lambda, d_err, sp);
        PARRM_HIP_CHECK(hipGetLastError());
        return PARRM_OK;
    }
    const size_t lds = (2 * 48 * 49 + 48 * 257) * sizeof(double);
    PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fit_solve_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(fit_solve_kernel, dim3(static_cast<unsigned>(n_periods)), dim3(256), lds, s, red, n, C, bw,
                       g.KP, g.nz, lambda, d_err);
    PARRM_HIP_CHECK(hipGetLastError());
    return PARRM_OK;
}

int parrm_fit_errors_host(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx, int64_t n_chans,
                          const double *h_periods, int64_t n_periods, int bw, double lambda, double *h_err,
                          void *d_workspace, size_t workspace_bytes, void *stream) {
    PARRM_REQUIRE(h_periods && h_err && d_workspace, "fit_errors_host: NULL argument");
    PARRM_REQUIRE(n_periods > 0 && n_periods <= 65535, "fit_errors_host: 1..65535 periods per call");
    const size_t inner = parrm_fit_workspace_bytes(n_idx, n_chans, n_periods, bw);
    PARRM_REQUIRE(inner != 0, "fit_errors_host: bad shape");
    const size_t extra = 2 * static_cast<size_t>(n_periods) * sizeof(double);
    if (workspace_bytes < inner + extra) {
        parrm::set_error("fit_errors_host: workspace too small (%zu < %zu)", workspace_bytes, inner + extra);
        return PARRM_ERR_WORKSPACE;
    }
    hipStream_t s = parrm::as_stream(stream);
    double *d_per = reinterpret_cast<double *>(static_cast<char *>(d_workspace) + inner);
    double *d_err = d_per + n_periods;
    int dev = 0;
    PARRM_HIP_CHECK(hipGetDevice(&dev));
    if (n_periods <= HostStage::kCap && !getenv("PARRM_FIT_COPY_PATH") && thread_stage().ready(dev)) {
        HostStage &st = thread_stage();
        // (<= kSmallPeriods: by value with the design-matrix kernel's launch; more: a device copy -- every workgroup of
        // that kernel reads them, which from mapped host memory would be one uncached PCIe read per wave)
        const bool by_value = n_periods <= kSmallPeriods && !getenv("PARRM_FIT_PERIODS_BY_COPY");
        if (!by_value) PARRM_HIP_CHECK(hipMemcpyAsync(d_per, h_periods, n_periods * sizeof(double), hipMemcpyHostToDevice, s));
        const unsigned long long seq = ++st.seq;
        volatile unsigned long long *h_flag = reinterpret_cast<volatile unsigned long long *>(st.h + 2 * HostStage::kCap);
        Publish pub{st.d + HostStage::kCap, reinterpret_cast<volatile unsigned long long *>(st.d + 2 * HostStage::kCap), seq,
                    false, by_value ? h_periods : nullptr};
        const int rc = fit_errors_impl(d_y, ldy, d_idx, n_idx, n_chans, d_per, n_periods, n_periods, bw, lambda, d_err,
                                       d_workspace, inner, stream, &pub);
        if (rc != PARRM_OK) return rc;
        if (!pub.done) {
            hipLaunchKernelGGL(fit_publish_kernel, dim3(1), dim3(64), 0, s, d_err, static_cast<int>(n_periods), pub.host_err,
                               pub.flag, seq);
            PARRM_HIP_CHECK(hipGetLastError());
        }
        // spin on the flag; look at the stream now and then so that a failed launch cannot hang us
        for (unsigned spins = 0; *h_flag != seq; ++spins) {
            if ((spins & 0xffff) == 0xffff) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) {
                    if (*h_flag == seq) break;
                    PARRM_HIP_CHECK(hipStreamSynchronize(s));
                    PARRM_REQUIRE(*h_flag == seq, "fit_errors_host: the stream drained without publishing the errors");
                    break;
                }
                if (q != hipErrorNotReady) return parrm::hip_fail(q, "fit_errors_host: stream");
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(h_err, st.h + HostStage::kCap, n_periods * sizeof(double));
        return PARRM_OK;
    }
    PARRM_HIP_CHECK(hipMemcpyAsync(d_per, h_periods, n_periods * sizeof(double), hipMemcpyHostToDevice, s));
    const int rc = parrm_fit_errors(d_y, ldy, d_idx, n_idx, n_chans, d_per, n_periods, bw, lambda, d_err, d_workspace,
                                    inner, stream);
    if (rc != PARRM_OK) return rc;
    PARRM_HIP_CHECK(hipMemcpyAsync(h_err, d_err, n_periods * sizeof(double), hipMemcpyDeviceToHost, s));
    PARRM_HIP_CHECK(hipStreamSynchronize(s));
    return PARRM_OK;
}

// Gang form of parrm_fit_errors_multi (see GangProblem): returns PARRM_OK with *handled = false when the problems do
// not all fit it (more than 256 channels, operands the matrix-core kernel cannot take, bandwidths without a fast
// solve) -- the caller then takes the one-stream-per-problem form.
static int fit_errors_gang(const parrm_fit_problem *problems, int n_problems, HostStage &st, hipStream_t s, bool *handled) {
    *handled = false;
    if (getenv("PARRM_FIT_MULTI_STREAMS") || !st.gang_ready()) return PARRM_OK;
    const char *accum_env = getenv("PARRM_FIT_ACCUM");
    if (accum_env && atoi(accum_env) == 1) return PARRM_OK;
    struct Shape {
        FitGeom g;
        size_t inner;
    };
    std::vector<Shape> shape(n_problems);
    for (int p = 0; p < n_problems; ++p) {
        const parrm_fit_problem &q = problems[p];
        const int K = 2 * q.bw + 1;
        if (q.n_chans > kYCols || !(K == 11 || K == 21 || K == 41)) return PARRM_OK;
        shape[p].g = fit_geometry(q.n_idx, q.n_chans, q.n_periods, q.bw);
        shape[p].inner = parrm_fit_workspace_bytes(q.n_idx, q.n_chans, q.n_periods, q.bw);
        const bool mfma_ok = q.ldy % 2 == 0 && (reinterpret_cast<uintptr_t>(q.d_y) & 15) == 0 &&
                             q.n_idx * q.ldy * 8 < 0x7fff0000LL && static_cast<int64_t>(shape[p].g.n_pad) * 48 * 8 * 3 < 0x7fff0000LL;
        if (!mfma_ok || q.n_idx >= (int64_t{1} << 30)) return PARRM_OK;
    }
    // host image: table | quarter flags | periods
    GangProblem *tab = reinterpret_cast<GangProblem *>(st.h_gang);
    int *quarters = reinterpret_cast<int *>(st.h_gang + HostStage::kMaxProblems * sizeof(GangProblem));
    double *h_per = reinterpret_cast<double *>(st.h_gang + HostStage::kMaxProblems * (sizeof(GangProblem) + sizeof(int)));
    const GangProblem *d_tab = reinterpret_cast<const GangProblem *>(st.d_gang);
    const int *d_quarters = reinterpret_cast<const int *>(st.d_gang + HostStage::kMaxProblems * sizeof(GangProblem));
    const double *d_per = reinterpret_cast<const double *>(st.d_gang + HostStage::kMaxProblems * (sizeof(GangProblem) + sizeof(int)));
    // Problems of one launch share bandwidth, kernel variant and -- roughly -- grid extents: a launch spans the
    // largest extents of its problems and every block outside a problem's own extents is dispatched just to return,
    // which is not free (a Nelder-Mead step of 10 candidates in 64 slices beside seven 330-candidate grids in 2
    // slices made 190 K empty workgroups: 33 ms).  Key: bandwidth, one-wave or four-wave Gram kernel, sample split,
    // power-of-two bucket of the candidate count.
    std::vector<int> order(n_problems);
    for (int p = 0; p < n_problems; ++p) order[p] = p;
    const bool no16 = getenv("PARRM_FIT_NO_NARROW16") != nullptr;
    auto width_class = [&](int64_t c) { return c <= 16 && !no16 ? 0 : (c <= 32 && !no16 ? 3 : (c <= 64 ? 1 : 2)); };
    auto bucket = [](int64_t v) {
        int b = 0;
        while ((int64_t{1} << b) < v) ++b;
        return b;
    };
    auto key = [&](int p) {
        return ((static_cast<int64_t>(problems[p].bw) * 4 + width_class(problems[p].n_chans)) * 128 + shape[p].g.nsplit) * 64 +
               bucket(problems[p].n_periods);
    };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
    std::vector<int64_t> err_off(n_problems);
    int64_t off = 0;
    for (int p = 0; p < n_problems; ++p) {
        err_off[p] = off;
        off += problems[p].n_periods;
    }
    for (int slot = 0; slot < n_problems; ++slot) {
        const int p = order[slot];
        const parrm_fit_problem &q = problems[p];
        const FitGeom &g = shape[p].g;
        double *wmat = static_cast<double *>(q.d_workspace);
        double *part = wmat + g.w_bytes / sizeof(double);
        double *d_err = reinterpret_cast<double *>(static_cast<char *>(q.d_workspace) + shape[p].inner) + q.n_periods;
        std::memcpy(h_per + err_off[p], q.h_periods, q.n_periods * sizeof(double));
        GangProblem &t = tab[slot];
        t.y = q.d_y;
        t.idx = q.d_idx;
        t.periods = d_per + err_off[p];
        t.wmat = wmat;
        t.part = part;
        t.red = g.nsplit > 1 ? part + g.part_bytes / sizeof(double) : part;
        t.err = d_err;
        t.ldy = q.ldy;
        t.lambda = q.lambda;
        t.n_idx = static_cast<int>(q.n_idx);
        t.n_pad = g.n_pad;
        t.n_chans = static_cast<int>(q.n_chans);
        t.n_periods = static_cast<int>(q.n_periods);
        t.nsplit = g.nsplit;
        const int ncb = 48 / g.KP;
        t.groups = g.packed ? g.groups : (t.n_periods + ncb - 1) / ncb;
        t.trig_blocks = (g.n_pad + 63) / 64;
        t.err_off = static_cast<int>(err_off[p]);
        quarters[slot] = (2 * q.bw + 1 == 41 && !getenv("PARRM_FIT_SOLVE_ONE_WORKGROUP")) ? g.KP : 0;
    }
    PARRM_HIP_CHECK(hipMemcpyAsync(st.d_gang, st.h_gang, HostStage::kGangBytes, hipMemcpyHostToDevice, s));
    const int exact = exact_trig_mode();
    for (int lo = 0; lo < n_problems;) {
        int hi = lo;
        while (hi < n_problems && key(order[hi]) == key(order[lo])) ++hi;
        const parrm_fit_problem &q0 = problems[order[lo]];
        const int bw = q0.bw, K = 2 * bw + 1, KP = kp_for(bw), ncb = 48 / KP;
        const int width = width_class(q0.n_chans);
        int max_trig = 0, max_groups = 0, max_split = 0, max_p = 0;
        bool any_split = false;
        for (int slot = lo; slot < hi; ++slot) {
            const GangProblem &t = tab[slot];
            max_trig = std::max(max_trig, t.trig_blocks);
            max_groups = std::max(max_groups, t.groups);
            max_split = std::max(max_split, t.nsplit);
            max_p = std::max(max_p, t.n_periods);
            any_split = any_split || t.nsplit > 1;
        }
        const unsigned nz = static_cast<unsigned>(hi - lo);
        const GangProblem *sub = d_tab + lo;
        const bool special = packed_form(bw);  // (one answer for the call: see fit_geometry)
        if (special)
            hipLaunchKernelGGL(fit_trig_packed_gang, dim3(max_trig, max_groups, nz), dim3(64 * packed_trig_waves(K)), 0, s, sub, bw, exact);
        else
            hipLaunchKernelGGL(fit_trig_stacked_gang, dim3(max_trig, max_groups, nz), dim3(64 * ncb), 0, s, sub, bw, KP, exact);
        const dim3 agrid(max_split, max_groups, nz);
#define PARRM_LAUNCH_GANG(NW_, CT_, THREADS_)                                                                           \
    do {                                                                                                               \
        if (special)                                                                                                   \
            hipLaunchKernelGGL((fit_accum_mfma_gang<NW_, CT_, 2>), agrid, dim3(THREADS_), 0, s, sub, KP, K);             \
        else                                                                                                           \
            hipLaunchKernelGGL((fit_accum_mfma_gang<NW_, CT_, 1>), agrid, dim3(THREADS_), 0, s, sub, KP, K);             \
    } while (0)
        if (width == 0)
            PARRM_LAUNCH_GANG(1, 1, 64);
        else if (width == 3)
            PARRM_LAUNCH_GANG(1, 2, 64);
        else if (width == 1)
            PARRM_LAUNCH_GANG(1, 4, 64);
        else
            PARRM_LAUNCH_GANG(4, 4, 256);
#undef PARRM_LAUNCH_GANG
        const int64_t elems = static_cast<int64_t>(KP + 1) * kNCol;
        const int sp = special ? 1 : 0;
        if (any_split)
            hipLaunchKernelGGL(fit_reduce_gang, dim3(static_cast<unsigned>((elems + 255) / 256), max_p, nz), dim3(256), 0, s, sub,
                               elems, special ? K : 0);
        if (K == 11)
            hipLaunchKernelGGL((fit_solve_fast_gang<11, 4>), dim3(max_p, 1, nz), dim3(256), 0, s, sub, KP, sp);
        else if (K == 21)
            hipLaunchKernelGGL((fit_solve_fast_gang<21, 4>), dim3(max_p, 1, nz), dim3(256), 0, s, sub, KP, sp);
        else if (quarters[lo] > 0)
            hipLaunchKernelGGL((fit_solve_fast_gang<41, 1>), dim3(max_p, 4, nz), dim3(64), 0, s, sub, KP, sp);
        else
            hipLaunchKernelGGL((fit_solve_fast_gang<41, 4>), dim3(max_p, 1, nz), dim3(256), 0, s, sub, KP, sp);
        PARRM_HIP_CHECK(hipGetLastError());
        lo = hi;
    }
    const unsigned long long seq = ++st.seq;
    volatile unsigned long long *h_flag = reinterpret_cast<volatile unsigned long long *>(st.h + 2 * HostStage::kCap);
    hipLaunchKernelGGL(fit_publish_gang, dim3(1), dim3(64), 0, s, d_tab, n_problems, d_quarters, st.d + HostStage::kCap,
                       reinterpret_cast<volatile unsigned long long *>(st.d + 2 * HostStage::kCap), seq);
    PARRM_HIP_CHECK(hipGetLastError());
    for (unsigned spins = 0; *h_flag != seq; ++spins) {
        if ((spins & 0xffff) == 0xffff) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {
                if (*h_flag == seq) break;
                PARRM_HIP_CHECK(hipStreamSynchronize(s));
                PARRM_REQUIRE(*h_flag == seq, "fit_errors_multi: the stream drained without publishing the errors");
                break;
            }
            if (e != hipErrorNotReady) return parrm::hip_fail(e, "fit_errors_multi: stream");
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int p = 0; p < n_problems; ++p)
        std::memcpy(problems[p].h_err, st.h + HostStage::kCap + err_off[p], problems[p].n_periods * sizeof(double));
    *handled = true;
    return PARRM_OK;
}

int parrm_fit_errors_multi(const parrm_fit_problem *problems, int n_problems, void *stream) {
    PARRM_REQUIRE(problems && n_problems > 0 && n_problems <= HostStage::kMaxProblems,
                  "fit_errors_multi: 1..%d problems per call", HostStage::kMaxProblems);
    int64_t total = 0;
    for (int p = 0; p < n_problems; ++p) {
        const parrm_fit_problem &q = problems[p];
        PARRM_REQUIRE(q.h_periods && q.h_err && q.d_workspace && q.n_periods > 0, "fit_errors_multi: problem %d: NULL or empty", p);
        const size_t inner = parrm_fit_workspace_bytes(q.n_idx, q.n_chans, q.n_periods, q.bw);
        PARRM_REQUIRE(inner != 0, "fit_errors_multi: problem %d: bad shape", p);
        if (q.workspace_bytes < inner + 2 * static_cast<size_t>(q.n_periods) * sizeof(double)) {
            parrm::set_error("fit_errors_multi: problem %d: workspace too small", p);
            return PARRM_ERR_WORKSPACE;
        }
        total += q.n_periods;
    }
    PARRM_REQUIRE(total <= HostStage::kCap, "fit_errors_multi: more than %lld candidates in one call", (long long)HostStage::kCap);
    int dev = 0;
    PARRM_HIP_CHECK(hipGetDevice(&dev));
    HostStage &st = thread_stage();
    if (!st.ready(dev) || !st.streams_ready()) {
        parrm::set_error("fit_errors_multi: cannot allocate the host hand-off block / side streams");
        return PARRM_ERR_HIP;
    }
    hipStream_t s = parrm::as_stream(stream);
    {
        bool handled = false;
        const int rc = fit_errors_gang(problems, n_problems, st, s, &handled);
        if (rc != PARRM_OK || handled) return rc;
    }
    const unsigned long long seq = ++st.seq;
    volatile unsigned long long *h_flags = reinterpret_cast<volatile unsigned long long *>(st.h + 2 * HostStage::kCap + 8);
    volatile unsigned long long *d_flags = reinterpret_cast<volatile unsigned long long *>(st.d + 2 * HostStage::kCap + 8);
    // fork: every side stream starts behind what is already queued on the caller's stream (the stage
    // matrices were written there)
    const int used = n_problems < HostStage::kStreams ? n_problems : HostStage::kStreams;
    PARRM_HIP_CHECK(hipEventRecord(st.fork, s));
    for (int k = 0; k < used; ++k) PARRM_HIP_CHECK(hipStreamWaitEvent(st.side[k], st.fork, 0));
    int64_t off = 0;
    for (int p = 0; p < n_problems; ++p) {
        const parrm_fit_problem &q = problems[p];
        hipStream_t sp = st.side[p % HostStage::kStreams];
        const size_t inner = parrm_fit_workspace_bytes(q.n_idx, q.n_chans, q.n_periods, q.bw);
        double *d_per = reinterpret_cast<double *>(static_cast<char *>(q.d_workspace) + inner);
        double *d_err = d_per + q.n_periods;
        std::memcpy(st.h + off, q.h_periods, q.n_periods * sizeof(double));  // page-locked staging: a true async copy
        PARRM_HIP_CHECK(hipMemcpyAsync(d_per, st.h + off, q.n_periods * sizeof(double), hipMemcpyHostToDevice, sp));
        const int rc = parrm_fit_errors(q.d_y, q.ldy, q.d_idx, q.n_idx, q.n_chans, d_per, q.n_periods, q.bw, q.lambda,
                                        d_err, q.d_workspace, inner, sp);
        if (rc != PARRM_OK) return rc;
        hipLaunchKernelGGL(fit_publish_kernel, dim3(1), dim3(64), 0, sp, d_err, static_cast<int>(q.n_periods),
                           st.d + HostStage::kCap + off, d_flags + p, seq);
        PARRM_HIP_CHECK(hipGetLastError());
        off += q.n_periods;
    }
    // join: later work on the caller's stream stays ordered behind the side streams
    for (int k = 0; k < used; ++k) {
        PARRM_HIP_CHECK(hipEventRecord(st.join[k], st.side[k]));
        PARRM_HIP_CHECK(hipStreamWaitEvent(s, st.join[k], 0));
    }
    off = 0;
    for (int p = 0; p < n_problems; ++p) {
        hipStream_t sp = st.side[p % HostStage::kStreams];
        for (unsigned spins = 0; h_flags[p] != seq; ++spins) {
            if ((spins & 0xffff) == 0xffff) {
                const hipError_t e = hipStreamQuery(sp);
                if (e == hipSuccess) {
                    if (h_flags[p] == seq) break;
                    PARRM_HIP_CHECK(hipStreamSynchronize(sp));
                    PARRM_REQUIRE(h_flags[p] == seq, "fit_errors_multi: a side stream drained without publishing its errors");
                    break;
                }
                if (e != hipErrorNotReady) return parrm::hip_fail(e, "fit_errors_multi: side stream");
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(problems[p].h_err, st.h + HostStage::kCap + off, problems[p].n_periods * sizeof(double));
        off += problems[p].n_periods;
    }
    return PARRM_OK;
}

int parrm_hip_shutdown(void) {
    // No call of this library may be in flight on another thread.  Streams the caller passed in are the
    // caller's; what is released here is what the library allocated for the life of the process.
    StageRegistry &r = registry();
    std::lock_guard<std::mutex> lock(r.mu);
    for (HostStage *st : r.all) st->release();  // hipHostFree waits for the device to be done with the block
    return PARRM_OK;
}

}  // extern "C"

// ---- host side of the device-carried Nelder-Mead chain (see NmChainState) ------------------------------------------
namespace parrm {

namespace {
struct ChainPlan {
    FitGeom widest;          // n_pad, KP, nz (the same for every batch size)
    size_t w_bytes = 0, part_bytes = 0, red_bytes = 0;
    int max_nsplit = 1, max_groups = 1;
    int nsplit_of[parrm_nmcore::kMaxBatch + 1], groups_of[parrm_nmcore::kMaxBatch + 1];
};
ChainPlan chain_plan(int64_t n_idx, int64_t n_chans, int bw) {
    ChainPlan c;
    c.nsplit_of[0] = 1;
    c.groups_of[0] = 0;
    for (int P = 1; P <= parrm_nmcore::kMaxBatch; ++P) {
        const FitGeom g = fit_geometry(n_idx, n_chans, P, bw);
        if (P == 1) c.widest = g;
        c.nsplit_of[P] = g.nsplit;
        c.groups_of[P] = g.groups;
        c.max_nsplit = std::max(c.max_nsplit, g.nsplit);
        c.max_groups = std::max(c.max_groups, g.groups);
        c.w_bytes = std::max(c.w_bytes, g.w_bytes);
        c.part_bytes = std::max(c.part_bytes, g.part_bytes);
        c.red_bytes = std::max(c.red_bytes, static_cast<size_t>(P) * g.nz * g.elems * sizeof(double));
    }
    return c;
}
constexpr size_t kChainStateBytes = (sizeof(NmChainState) + 255) / 256 * 256;
constexpr size_t kChainTableBytes = static_cast<size_t>(kChainTable) * 16;
bool chain_shape_ok(int64_t n_idx, int64_t n_chans, int bw) {
    const int K = 2 * bw + 1;
    return n_idx > 0 && n_idx < (int64_t{1} << 30) && n_chans > 0 && n_chans < (int64_t{1} << 24) && bw >= 0 && bw <= kMaxBw &&
           (K == 11 || K == 21 || K == 41) && packed_form(bw);
}
}  // namespace

size_t nm_chain_workspace_bytes(int64_t n_idx, int64_t n_chans, int bw) {
    if (!chain_shape_ok(n_idx, n_chans, bw)) return 0;
    const ChainPlan c = chain_plan(n_idx, n_chans, bw);
    return c.w_bytes + c.part_bytes + c.red_bytes + 256 * sizeof(double) + kChainStateBytes + kChainTableBytes + 512;
}

// Runs the refinement `init` describes (a fresh parrm_nmcore::Core: no step taken, nothing known) to its end on the
// device.  *handled = false (with PARRM_OK): the shape is not one the chain takes, or the device gave up (history or
// table full) -- the caller steps the refinement from the host instead.  On success the recorded batches are in
// hist_x / hist_f / batch_sizes and the runs' results in res_*.
int nm_chain_run(const parrm_nmcore::Core &init, const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                 int64_t n_chans, int bw, double lambda, void *d_workspace, size_t workspace_bytes, void *stream, double *hist_x,
                 double *hist_f, int hist_capacity, int *batch_sizes, int batch_capacity, int *n_batches, int *hist_used,
                 double *res_x, double *res_f, int *res_its, int *res_calls, bool *handled) {
    *handled = false;
    // Opt-in (PARRM_NM_CHAIN=1).  Built in round 4 to take the host out of the loop between two optimiser batches and
    // measured in the bench step against the host-stepped loop (profiles/r04_nm_chain_ab.txt): the same kernels, the
    // same bits, and 0.65 ms SLOWER per find_period -- the step kernel that replaces the hand-off takes 10-15 us (a
    // launch, two dependent trips to L2 for the state and the errors, the state machine, the write-back) against the
    // 12-16 us the host needs to see the flag, decide and launch; each refinement also pays ~50 us of set-up and two
    // batch groups that drain as no-ops.  The hand-off was not the lever: the chain's time is its kernels.
    const char *chain_env = getenv("PARRM_NM_CHAIN");
    if (!chain_env || atoi(chain_env) == 0 || getenv("PARRM_NM_HOST_STEPPED") || !chain_shape_ok(n_idx, n_chans, bw)) return PARRM_OK;
    if (init.n_runs < 1 || init.n_runs > parrm_nmcore::kMaxRuns || exact_trig_mode()) return PARRM_OK;
    const char *accum_env = getenv("PARRM_FIT_ACCUM");
    if (accum_env && atoi(accum_env) == 1) return PARRM_OK;
    const ChainPlan plan = chain_plan(n_idx, n_chans, bw);
    const FitGeom &g = plan.widest;
    const bool mfma_ok = ldy % 2 == 0 && (reinterpret_cast<uintptr_t>(d_y) & 15) == 0 && n_idx * ldy * 8 < 0x7fff0000LL &&
                         static_cast<int64_t>(g.n_pad) * 48 * 8 * 3 < 0x7fff0000LL;
    if (!mfma_ok || static_cast<int64_t>(parrm_nmcore::kMaxBatch) * g.nz > 65535) return PARRM_OK;
    const size_t need = nm_chain_workspace_bytes(n_idx, n_chans, bw);
    if (workspace_bytes < need) return PARRM_OK;
    int dev = 0;
    PARRM_HIP_CHECK(hipGetDevice(&dev));
    HostStage &hs = thread_stage();
    if (!hs.ready(dev) || !hs.chain_ready()) return PARRM_OK;
    hipStream_t s = as_stream(stream);

    // workspace: design matrices | partial blocks | reduced blocks | errors | state | table
    char *ws = static_cast<char *>(d_workspace);
    double *wmat = reinterpret_cast<double *>(ws);
    double *part = reinterpret_cast<double *>(ws + plan.w_bytes);
    double *red = reinterpret_cast<double *>(ws + plan.w_bytes + plan.part_bytes);
    double *d_err = reinterpret_cast<double *>(ws + plan.w_bytes + plan.part_bytes + plan.red_bytes);
    char *tail = ws + plan.w_bytes + plan.part_bytes + plan.red_bytes + 256 * sizeof(double);
    tail = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(tail) + 255) & ~uintptr_t{255});
    NmChainState *d_state = reinterpret_cast<NmChainState *>(tail);
    unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(tail + kChainStateBytes);
    double *d_vals = reinterpret_cast<double *>(tail + kChainStateBytes + static_cast<size_t>(kChainTable) * 8);

    NmChainHost *h = hs.h_chain;
    const long long run_id = ++hs.chain_runs;
    h->status = 0;
    NmChainState &st0 = h->init;
    std::memset(&st0, 0, sizeof st0);
    st0.core = init;
    st0.status = kChainRunning;
    st0.nsplit = 1;
    std::memcpy(st0.nsplit_of, plan.nsplit_of, sizeof plan.nsplit_of);
    std::memcpy(st0.groups_of, plan.groups_of, sizeof plan.groups_of);
    std::atomic_thread_fence(std::memory_order_release);  // (the first step kernel reads the block through its device mapping)

    const int n = static_cast<int>(n_idx), C = static_cast<int>(n_chans), K = 2 * bw + 1;
    const bool wide = C > 64 && !(C <= 128 && g.nz == 1 && !getenv("PARRM_FIT_NO_TWO_WAVES"));
    const bool fused = wide && !getenv("PARRM_FIT_UNFUSED");
    const bool narrow_ok = !getenv("PARRM_FIT_NO_NARROW16");
    const bool quarters = K == 41 && 4 * g.nz <= 65535 && !getenv("PARRM_FIT_SOLVE_ONE_WORKGROUP");
    if (fused) {
        static std::atomic<bool> lds_allowed[64];
        if (dev < 0 || dev >= 64 || !lds_allowed[dev].load(std::memory_order_acquire)) {
            PARRM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nm_chain_fused_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLdsBytes));
            if (dev >= 0 && dev < 64) lds_allowed[dev].store(true, std::memory_order_release);
        }
    }
    // The largest batch THIS refinement can form: 2 abscissae per run at the start, 3 per run in flight afterwards, plus
    // 6 of look-ahead for each of at most `lookahead_runs` runs (expansion / shrink follow-ups are smaller).  The grids
    // are cut for it, not for kMaxBatch: blocks beyond a batch's geometry return at once, but dispatching them is not
    // free (measured with grids for 32 candidates: +18 us per Gram launch of 9).
    const int la = std::max(0, std::min(init.n_runs, init.lookahead_runs));
    const int Pmax = std::min(parrm_nmcore::kMaxBatch, std::max(3 * init.n_runs, 9 * la));
    int max_nsplit = 1, max_groups = 1, max_blocks = 1;
    for (int P = 1; P <= Pmax; ++P) {
        max_nsplit = std::max(max_nsplit, plan.nsplit_of[P]);
        max_groups = std::max(max_groups, plan.groups_of[P]);
        max_blocks = std::max(max_blocks, plan.nsplit_of[P] * plan.groups_of[P]);
    }
    const dim3 gram_grid(max_blocks, 1, g.nz);  // (see chain_gram_block)
    unsigned long long seq = hs.seq;  // (the progress word only ever grows: shared numbering with the publish flag's)
    auto launch_step = [&](int first) {
        ++seq;
        hipLaunchKernelGGL(nm_chain_step_kernel, dim3(1), dim3(64), 0, s, d_state, d_keys, d_vals, part, red, d_err, g.KP, g.nz, C,
                           quarters ? 1 : 0, hs.d_chain, seq, run_id, first, Pmax);
        return hipGetLastError();
    };
    auto launch_batch = [&]() -> hipError_t {
        if (fused) {
            hipLaunchKernelGGL(nm_chain_fused_kernel, gram_grid, dim3(256), kFusedLdsBytes, s, d_state, d_y, ldy, d_idx, g.n_pad, n,
                               C, g.KP, part, K);
        } else {
            hipLaunchKernelGGL(nm_chain_trig_kernel, dim3((g.n_pad + 63) / 64, max_groups), dim3(64 * packed_trig_waves(K)), 0,
                               s, d_state, d_idx, n, g.n_pad, bw, wmat);
            if (C <= 16 && narrow_ok)
                hipLaunchKernelGGL((nm_chain_accum_kernel<1, 1>), gram_grid, dim3(64), 0, s, d_state, d_y, ldy, wmat, g.n_pad, n, C,
                                   g.KP, part, K);
            else if (C <= 32 && narrow_ok)
                hipLaunchKernelGGL((nm_chain_accum_kernel<1, 2>), gram_grid, dim3(64), 0, s, d_state, d_y, ldy, wmat, g.n_pad, n, C,
                                   g.KP, part, K);
            else if (C <= 64)
                hipLaunchKernelGGL((nm_chain_accum_kernel<1, 4>), gram_grid, dim3(64), 0, s, d_state, d_y, ldy, wmat, g.n_pad, n, C,
                                   g.KP, part, K);
            else if (C <= 128 && g.nz == 1 && !getenv("PARRM_FIT_NO_TWO_WAVES"))
                hipLaunchKernelGGL((nm_chain_accum_kernel<2, 4>), gram_grid, dim3(128), 0, s, d_state, d_y, ldy, wmat, g.n_pad, n,
                                   C, g.KP, part, K);
            else
                hipLaunchKernelGGL((nm_chain_accum_kernel<4, 4>), gram_grid, dim3(256), 0, s, d_state, d_y, ldy, wmat, g.n_pad, n,
                                   C, g.KP, part, K);
        }
        if (max_nsplit > 1)
            hipLaunchKernelGGL(nm_chain_reduce_kernel, dim3(static_cast<unsigned>((g.elems + 255) / 256), Pmax * g.nz), dim3(256), 0,
                               s, d_state, part, g.elems, g.nz, C, red, K);
        if (quarters) {
            hipLaunchKernelGGL((nm_chain_solve_kernel<41, 1>), dim3(Pmax, 4 * g.nz), dim3(64), 0, s, d_state, part, red, n, C, g.KP,
                               g.nz, lambda, d_err);
        } else if (K == 41) {
            hipLaunchKernelGGL((nm_chain_solve_kernel<41, 4>), dim3(Pmax), dim3(256), 0, s, d_state, part, red, n, C, g.KP, g.nz,
                               lambda, d_err);
        } else if (K == 21) {
            hipLaunchKernelGGL((nm_chain_solve_kernel<21, 4>), dim3(Pmax), dim3(256), 0, s, d_state, part, red, n, C, g.KP, g.nz,
                               lambda, d_err);
        } else {
            hipLaunchKernelGGL((nm_chain_solve_kernel<11, 4>), dim3(Pmax), dim3(256), 0, s, d_state, part, red, n, C, g.KP, g.nz,
                               lambda, d_err);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_step(0);
    };
    // batches queued beyond the one the device is working on: the host's ~15 us per batch (4 launches) must stay ahead
    // of the device's 50-170 us; what is still queued when the refinement ends drains in ~3 us per kernel
    int ahead = 2;
    if (const char *env = getenv("PARRM_NM_CHAIN_AHEAD")) ahead = std::max(1, std::min(16, atoi(env)));
    const unsigned long long base = seq;
    hipError_t e = launch_step(1);
    if (e != hipSuccess) {
        hs.seq = seq;
        return hip_fail(e, "nm_chain: launch");
    }
    // Batch group k = the kernels of batch k and the step kernel that closes it and forms batch k + 1; everything is
    // stream-ordered, so groups are queued before the state they will read exists.
    long long queued = 0;
    unsigned spins = 0;
    int rc = PARRM_OK;
    while (true) {
        if ((h->status >> 8) == run_id) break;
        const unsigned long long progress = h->progress;
        const long long steps_done = progress > base ? static_cast<long long>(progress - base) : 0;  // step kernels that ran
        if (queued < steps_done + ahead) {
            e = launch_batch();
            if (e != hipSuccess) {
                rc = hip_fail(e, "nm_chain: launch");
                break;
            }
            ++queued;
            continue;
        }
        if ((++spins & 0xffff) == 0) {  // look at the stream now and then so that a failed launch cannot hang us
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) {
                rc = hip_fail(q, "nm_chain: stream");
                break;
            }
            if (q == hipSuccess) {  // every queued step kernel has run: its words are visible now or never
                if ((h->status >> 8) == run_id) break;
                if (h->progress == progress) {
                    set_error("nm_chain: the stream drained without progress");
                    rc = PARRM_ERR_HIP;
                    break;
                }
            }
        }
        __builtin_ia32_pause();
    }
    hs.seq = seq;
    if (rc != PARRM_OK) return rc;
    std::atomic_thread_fence(std::memory_order_acquire);
    if ((h->status & 0xff) != kChainFinished) return PARRM_OK;  // the device gave up: *handled stays false
    const int nb = h->n_batches, used = h->hist_used;
    if (nb > batch_capacity || used > hist_capacity) return PARRM_OK;
    std::memcpy(hist_x, const_cast<const double *>(h->hist_x), static_cast<size_t>(used) * sizeof(double));
    std::memcpy(hist_f, const_cast<const double *>(h->hist_f), static_cast<size_t>(used) * sizeof(double));
    std::memcpy(batch_sizes, const_cast<const int *>(h->batch_sizes), static_cast<size_t>(nb) * sizeof(int));
    for (int r = 0; r < init.n_runs; ++r) {
        res_x[r] = h->res_x[r];
        res_f[r] = h->res_f[r];
        res_its[r] = h->res_its[r];
        res_calls[r] = h->res_calls[r];
    }
    *n_batches = nb;
    *hist_used = used;
    *handled = true;
    return PARRM_OK;
}

}  // namespace parrm
