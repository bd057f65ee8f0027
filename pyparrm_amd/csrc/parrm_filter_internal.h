// Internal to the filter_data kernels (parrm_filter.hip, parrm_filter_phase.hip).
#pragma once

#include <vector>

#include <atomic>
#include <memory>
#include <thread>

#include "parrm_common.h"

namespace parrm_filter {

struct FilterArgs {
    const void *x;
    void *y;
    int64_t n_chans;
    int64_t plan_chans;          // channels the stretch geometry is planned for (>= n_chans: a channel block of a sharded recording is cut like the whole recording, so every channel's result is bit-identical)
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
    int32_t w_pos_min, w_neg_min;  // smallest |w| among the taps with w > 0 / w < 0 (0: the filter has none of that sign)
};

// number of taps whose source sample n-w lies inside [0, n_total)
__device__ inline int valid_taps(const FilterArgs &a, int64_t n) {
    const int64_t hw = a.hw;
    const int64_t w_hi = n < hw ? n : hw;
    int64_t w_lo = n - a.n_total + 1;
    if (w_lo < -hw) w_lo = -hw;
    if (w_hi < w_lo) return 0;
    return a.tapcum[w_hi + hw + 1] - a.tapcum[w_lo + hw];
}

// ZERO_NONFINITE: apply parrm.py:869 here (gather / repair evaluation).  The recurrence kernels pass false
// and store non-finite results as they are: filter_repair_kernel finds them and recomputes (see below).
template <typename TO, bool ZERO_NONFINITE = true>
__device__ inline void emit(const FilterArgs &a, int64_t c, int64_t n, double xc, double s) {
    double y;
    if (n >= a.hw && n + a.hw < a.n_total) {
        y = xc - s * a.inv_taps;
    } else {
        const int v = valid_taps(a, n);
        // no tap inside the recording: 0 (the reference's 0/0 -> NaN -> 0).  The recurrence kernels keep a poisoned
        // running sum VISIBLE there (s * 0 is NaN for a non-finite s, 0 otherwise): with a one-sided filter the last
        // outputs of a recording have no valid tap, and they are where the repair pass looks for the poison
        // (scripts/fuzz_filter_r2.py, seed 32 case 138)
        y = v > 0 ? xc - s / static_cast<double>(v) : (ZERO_NONFINITE ? 0.0 : s * 0.0);
    }
    if (ZERO_NONFINITE && !isfinite(y)) y = 0.0;  // parrm.py:869
    static_cast<TO *>(a.y)[c * a.ldy + (n - a.out_first)] = static_cast<TO>(y);
}

// direct evaluation of one output from global memory (the gather kernel's arithmetic)
template <typename TI, typename TO>
__device__ inline void gather_output(const FilterArgs &a, int64_t c, int64_t n) {
    const TI *row = static_cast<const TI *>(a.x) + c * a.ldx - a.buf_first;
    double s = 0.0;
    for (int r = 0; r < a.n_runs; ++r) {
        int64_t lo = n - a.runs[2 * r + 1];
        int64_t hi = n - a.runs[2 * r];
        if (lo < 0) lo = 0;
        if (hi > a.n_total - 1) hi = a.n_total - 1;
        for (int64_t g = lo; g <= hi; ++g) s += static_cast<double>(row[g]);
    }
    emit<TO>(a, c, n, static_cast<double>(row[n]), s);
}

// Non-finite inputs and the recurrence kernels.  Those kernels carry a running tap sum per residue
// class of a (channel, stretch); a NaN/Inf sample that enters it never leaves (NaN - NaN, Inf - Inf),
// so from there to the end of the stretch every output of that class is non-finite -- although only the
// outputs whose taps reach the bad sample should be 0 (parrm.py:869) and the others have finite values.
// The recurrence kernels therefore do NOT apply :869 themselves: they store what they computed, and this
// pass runs behind every recurrence launch.
//   (1) Because the poison survives to the end of a stretch, a stretch in which a bad sample was a TAP of
//       any output shows a non-finite value among its last q <= 512 outputs (each residue class has one
//       there); the pass probes those outputs of every stretch (0.1 % of the output, L2-warm) and
//       recomputes a stretch that shows one tap by tap, as the gather kernel does, zeroing exactly the
//       outputs whose taps reach a non-finite sample.
//   (2) A bad sample x[p] that is a tap of NO output of its stretch spoils exactly one output there, its
//       own (y[p] = x[p] - mean), and leaves no trace at the tail.  x[p] is a tap of the outputs p + w, so it
//       is a tap of none in [o0, o1) iff p + w_pos_min >= o1 (or the filter has no tap with w > 0) AND
//       p - w_neg_min < o0 (or no tap with w < 0): for a two-sided filter nowhere (unless the stretch is
//       shorter than w_pos_min + w_neg_min), for a one-sided one the first or last few samples of every
//       stretch -- a "past" filter never reads x[p] for an output at or behind p.  The pass reads the input
//       over that interval and stores 0 wherever the sample itself is non-finite, which is what :869 makes
//       of such an output whatever its taps hold.  (Found by scripts/fuzz_filter_r2.py, case 182.  Testing
//       every loaded sample inside the phase kernel instead cost it 2 % (float64) to 4.5 % (float32): one
//       more vector instruction per output at its issue limit.)
// Clean recordings pay the probe (and, with a one-sided filter, that short read) only, the hot loops lose the per-output finiteness
// test (three vector instructions per output), and an all-zero channel (a dead electrode) is not mistaken
// for a poisoned one.  A flag kept in the phase kernel cost it a VGPR -- one wave per SIMD at its
// 80-register budget (9.5 -> 10.8 ms).  Result: every kernel variant and every chunking returns what the
// direct evaluation returns, on the poisoned channel too.
constexpr int kRepairStretchesPerBlock = 4;
constexpr int kRepairProbe = 512;

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) filter_repair_kernel(FilterArgs a) {
    const int64_t n_str = a.n_chans * a.n_stretch;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRepairStretchesPerBlock;
    TO *y = static_cast<TO *>(a.y);
    for (int64_t s = first; s < first + kRepairStretchesPerBlock && s < n_str; ++s) {
        const int64_t c = s / a.n_stretch, st = s - c * a.n_stretch;
        const int64_t o0 = st * a.stretch_len;
        const int64_t o1 = o0 + a.stretch_len < a.out_len ? o0 + a.stretch_len : a.out_len;
        const int64_t p0 = o1 - kRepairProbe > o0 ? o1 - kRepairProbe : o0;
        int hit = 0;
        for (int64_t i = p0 + threadIdx.x; i < o1; i += blockDim.x)
            if (!isfinite(static_cast<double>(y[c * a.ldy + i]))) hit = 1;
        if (__syncthreads_or(hit)) {  // workgroup-uniform
            for (int64_t i = o0 + threadIdx.x; i < o1; i += blockDim.x) gather_output<TI, TO>(a, c, a.out_first + i);
            continue;
        }
        // (2): samples of this stretch that are taps of none of its outputs
        const int64_t lo = a.w_pos_min > 0 && o1 - a.w_pos_min > o0 ? o1 - a.w_pos_min : o0;
        const int64_t hi = a.w_neg_min > 0 && o0 + a.w_neg_min < o1 ? o0 + a.w_neg_min : o1;
        const TI *row = static_cast<const TI *>(a.x) + c * a.ldx - a.buf_first + a.out_first;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
            if (!isfinite(static_cast<double>(row[i]))) y[c * a.ldy + i] = TO(0);
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
    // packed float32 form only (parrm_filter_plan::phase_pack): two ring copies, one 8-byte read per tap
    int32_t copy_o_bytes; // byte offset of copy O (row slot s at element s + 1) from copy E; rs is even, rs/2 odd
    int32_t tab_off;      // offset (in int32) of this geometry's table within the plan's phase table
    const int32_t *tab;   // [M][2*d_pad] tap offsets per table row; [3*n_runs] (a', b_lo, b_hi); wrap: [2*d_pad] b
};

}  // namespace parrm_filter

namespace parrm_filter {
// parrm_filter_comb.hip: the per-filter generated kernel (float64 recordings, comb filters with q in [80, 176])
struct CombKernel;
struct CombJob {  // one background build (compile, load, spill check, self-test) of a plan's generated kernel
    std::thread worker;
    std::atomic<int> done{0};  // set last by the worker: result_state / kernel are final
    int result_state = -1;     // 1 loaded and verified, -1 unavailable
    CombKernel *kernel = nullptr;
};
CombKernel *comb_generate(const std::vector<int8_t> &tap, int64_t hw, int64_t q, int attempt = 0, bool in32 = false,
                          bool out32 = false);
void comb_destroy(CombKernel *k);
bool comb_load(CombKernel *k);  // code object from the caches or hipRTC, loaded on the current device
bool comb_code_cached(const CombKernel *k);  // a code object for its source is in the in-tree directory or the user cache
const char *comb_error(const CombKernel *k);
void comb_set_error(CombKernel *k, const char *msg);
int comb_reach(const CombKernel *k);
double comb_reads_per_output(const CombKernel *k);
int comb_stride(const CombKernel *k);
bool comb_accepts(const CombKernel *k, const FilterArgs &a);
int launch_comb(const CombKernel *k, FilterArgs *a, hipStream_t stream);
int64_t comb_search_stride(const std::vector<int8_t> &tap, int64_t hw);
}  // namespace parrm_filter

struct parrm_filter_plan {
    int device = 0;
    int64_t hw = 0, n_taps = 0, n_runs = 0;
    int64_t w_pos_min = 0, w_neg_min = 0;  // nearest tap behind / ahead of the centre (0: none on that side)
    int64_t q = 0, n_delta = 0, n_delta_pad = 0;
    int ring_log2_f64 = 0, rows_per_fill = 0, block_threads = 0;
    int forced_kernel = PARRM_KERNEL_AUTO;
    int32_t *d_tables = nullptr;  // runs | tapcum | delta offsets, one allocation
    double *d_weights = nullptr;  // delta weights
    int64_t off_tapcum = 0, off_delta = 0;
    // phase-major kernel (0 groups = not available for this filter)
    parrm_filter::PhaseGeom phase{};
    // packed float32 form of the guarded plan (own shape and ring: two copies must fit; n_groups == 0: none);
    // its table holds BYTE offsets and lives behind the main one in d_phase_tab
    parrm_filter::PhaseGeom phase_pack{};
    int32_t *d_phase_tab = nullptr;
    // Segmented form (half-widths beyond any LDS ring; parrm_filter.hip): the taps cut into windows of
    // offsets, each a plan of its own whose phase kernel adds its raw tap sums into a float64 accumulator.
    // seg_centre[k] is the offset w the k-th sub-plan's centre stands for.  Empty: not segmented.
    std::vector<parrm_filter_plan *> segments;
    std::vector<int64_t> seg_centre;
    // Generated kernel (parrm_filter_comb.hip).  The tap mask is kept so that the kernel can be generated at the
    // first launch large enough to pay for a compile; comb_state: 0 not tried, 1 loaded, -1 unavailable.
    std::vector<int8_t> tap_mask;
    // One kernel per pair of element types (kCombVariants: 0 = float64 -> float64, 1 = float32 -> float64,
    // 2 = float32 -> float32); comb_last = the variant the last launch asked for (parrm_filter_plan_generated).
    mutable parrm_filter::CombKernel *comb[3] = {nullptr, nullptr, nullptr};
    mutable int comb_state[3] = {0, 0, 0};
    mutable int comb_last = 0;
    // comb_state 2 = being compiled by a worker thread (parrm_filter_plan_set_background): launches run the generic
    // kernels meanwhile; the job's result is adopted by the next launch or query after it has finished
    bool comb_background = false;
    mutable std::shared_ptr<parrm_filter::CombJob> comb_job[3];
};


namespace parrm_filter {
// parrm_filter_phase.hip
void plan_phase(const std::vector<int8_t> &tap, int64_t hw, parrm_filter_plan *plan, std::vector<int32_t> *table);
// (fills a->stretch_len / a->n_stretch: the repair pass needs the stretch geometry of the launch)
// mode 0: y = x - mean of taps; TO = double only: mode 1: y += raw tap sum (a segment of a long filter), mode 2: y = it
template <typename TI, typename TO>
int launch_phase(const parrm_filter_plan *plan, FilterArgs *a, hipStream_t stream, int mode = 0);
}  // namespace parrm_filter
