// Internal to the filter_data kernels (parrm_filter.hip, parrm_filter_phase.hip).
#pragma once

#include <vector>

#include "parrm_common.h"

namespace parrm_filter {

struct FilterArgs {
    const void *x;
    void *y;
    int64_t n_chans;
    int64_t buf_first, buf_len;  // samples [buf_first, buf_first+buf_len) are addressable in x
    int64_t out_first, out_len;  // outputs to produce
    int64_t n_total;             // length of the whole recording (edge logic)
    int64_t ldx, ldy;
    const int32_t *runs;    // [n_runs][2] inclusive tap runs (w_lo, w_hi), ascending
    const int32_t *tapcum;  // [2*hw+2]: tapcum[j] = #taps with w < j - hw
    const int32_t *delta;   // [n_delta_pad] offsets u of d_q = tap(.+q) - tap(.)
    const double *delta_w;  // [n_delta_pad] +1 / -1, and 0 for the padding entries
    int32_t n_runs, n_delta_pad;
    int32_t hw, n_taps;
    int32_t q, ring_mask, rows_per_fill;
    int64_t stretch_len, n_stretch;
    double inv_taps;
    // Non-finite inputs.  The recurrence kernels carry a running sum per residue class; a NaN/Inf sample
    // that enters it never leaves (NaN - NaN), so a workgroup that ever produces a non-finite result
    // marks the 64 K-output tiles of its stretch here and filter_repair_kernel recomputes those tiles
    // tap by tap: outputs whose taps reach a non-finite sample are 0 (parrm.py:869), every other output
    // is exact -- the same answer from every kernel variant and every chunking.  NULL = not tracked
    // (the gather kernel, which is exact by construction).
    unsigned int *poison;      // [n_chans][poison_tiles]
    int64_t poison_tiles;      // tiles per channel
};

constexpr int64_t kPoisonTile = 65536;
constexpr int kRepairTilesPerBlock = 16;

// number of taps whose source sample n-w lies inside [0, n_total)
__device__ inline int valid_taps(const FilterArgs &a, int64_t n) {
    const int64_t hw = a.hw;
    const int64_t w_hi = n < hw ? n : hw;
    int64_t w_lo = n - a.n_total + 1;
    if (w_lo < -hw) w_lo = -hw;
    if (w_hi < w_lo) return 0;
    return a.tapcum[w_hi + hw + 1] - a.tapcum[w_lo + hw];
}

// returns true when the result was not finite (and 0 was stored, parrm.py:869)
template <typename TO>
__device__ inline bool emit(const FilterArgs &a, int64_t c, int64_t n, double xc, double s) {
    double y;
    if (n >= a.hw && n + a.hw < a.n_total) {
        y = xc - s * a.inv_taps;
    } else {
        const int v = valid_taps(a, n);
        y = v > 0 ? xc - s / static_cast<double>(v) : 0.0;
    }
    const bool bad = !isfinite(y);
    if (bad) y = 0.0;  // parrm.py:869
    static_cast<TO *>(a.y)[c * a.ldy + (n - a.out_first)] = static_cast<TO>(y);
    return bad;
}

// end of a recurrence workgroup: mark the tiles of outputs [s0, s1) of channel c when any thread saw a
// non-finite result (every thread of the workgroup calls this)
__device__ inline void mark_poison(const FilterArgs &a, bool bad, int64_t c, int64_t s0, int64_t s1) {
    if (a.poison == nullptr) return;
    if (!__syncthreads_or(bad ? 1 : 0)) return;
    const int64_t t0 = (s0 - a.out_first) / kPoisonTile, t1 = (s1 - 1 - a.out_first) / kPoisonTile;
    for (int64_t t = t0 + threadIdx.x; t <= t1; t += blockDim.x) a.poison[c * a.poison_tiles + t] = 1u;
}

// direct evaluation of one output from global memory (the gather kernel's arithmetic)
template <typename TI, typename TO>
__device__ inline bool gather_output(const FilterArgs &a, int64_t c, int64_t n) {
    const TI *row = static_cast<const TI *>(a.x) + c * a.ldx - a.buf_first;
    double s = 0.0;
    for (int r = 0; r < a.n_runs; ++r) {
        int64_t lo = n - a.runs[2 * r + 1];
        int64_t hi = n - a.runs[2 * r];
        if (lo < 0) lo = 0;
        if (hi > a.n_total - 1) hi = a.n_total - 1;
        for (int64_t g = lo; g <= hi; ++g) s += static_cast<double>(row[g]);
    }
    return emit<TO>(a, c, n, static_cast<double>(row[n]), s);
}

// One workgroup looks at kRepairTilesPerBlock tile flags and recomputes the marked tiles.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) filter_repair_kernel(FilterArgs a) {
    const int64_t n_tiles = a.n_chans * a.poison_tiles;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRepairTilesPerBlock;
    for (int64_t t = first; t < first + kRepairTilesPerBlock && t < n_tiles; ++t) {
        if (a.poison[t] == 0u) continue;  // workgroup-uniform
        const int64_t c = t / a.poison_tiles, tile = t - c * a.poison_tiles;
        const int64_t lo = tile * kPoisonTile;
        const int64_t hi = lo + kPoisonTile < a.out_len ? lo + kPoisonTile : a.out_len;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) gather_output<TI, TO>(a, c, a.out_first + i);
    }
}

template <typename TI>
__device__ inline TI load_padded(const FilterArgs &a, const TI *row, int64_t g) {
    // zero outside the recording; the window contract guarantees everything else is addressable
    const bool ok = g >= 0 && g < a.n_total && g >= a.buf_first && g < a.buf_first + a.buf_len;
    return ok ? row[g - a.buf_first] : TI(0);
}


// ---------------------------------------------------------------- phase-major kernel geometry
struct PhaseGeom {
    int32_t q, qp;        // stride; q rounded up to whole waves (threads per row group)
    int32_t guard;        // B: residues mirrored on each side of a row
    int32_t m_slots;      // M: row slots of the ring
    int32_t rs;           // elements per residue: M + R - 1 (mirrored head), made odd
    int32_t n_groups;     // NG row groups sharing the ring
    int32_t rows;         // R rows per thread per iteration (template parameter of the launch)
    int32_t a_lo, a_hi;   // tap rows reach from m - a_hi to m - a_lo (+1 look-ahead)
    int32_t d_pad;        // padded length of the +1 and of the -1 delta lists
    int32_t n_runs;       // full-tap runs for the per-stretch initialisation
    int32_t debug;        // ablation bits for profiling builds (PARRM_DEBUG_FLAGS); 0 in production
    int32_t wrap;         // 1: residues b in [0, q) with per-lane wrap to the previous row (guard == 0)
    const int32_t *tab;   // [M][2*d_pad] tap offsets per table row; [3*n_runs] (a', b_lo, b_hi); wrap: [2*d_pad] b
};

}  // namespace parrm_filter

struct parrm_filter_plan {
    int device = 0;
    int64_t hw = 0, n_taps = 0, n_runs = 0;
    int64_t q = 0, n_delta = 0, n_delta_pad = 0;
    int ring_log2_f64 = 0, rows_per_fill = 0, block_threads = 0;
    int forced_kernel = PARRM_KERNEL_AUTO;
    int32_t *d_tables = nullptr;  // runs | tapcum | delta offsets, one allocation
    double *d_weights = nullptr;  // delta weights
    int64_t off_tapcum = 0, off_delta = 0;
    // phase-major kernel (0 groups = not available for this filter)
    parrm_filter::PhaseGeom phase{};
    int32_t *d_phase_tab = nullptr;
};


namespace parrm_filter {
// parrm_filter_phase.hip
void plan_phase(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *plan, std::vector<int32_t> *table);
template <typename TI, typename TO>
int launch_phase(const parrm_filter_plan *plan, FilterArgs a, hipStream_t stream);
}  // namespace parrm_filter
