// filter_data, per-filter GENERATED kernel ("comb" kernel) -- parrm.py:861-869 for float64 recordings.
//
// Same recurrence as the phase-major kernel (parrm_filter_phase_impl.h),
//
//     S(n+q) = S(n) + sum_u d_q(u) xz[n-u],        d_q(u) = tap(u+q) - tap(u),     u = a*q + b, |b| <= HB,
//
// but the tap geometry is COMPILE-TIME data of a kernel generated for the one filter (hipRTC at first use,
// code objects cached on disk and shipped in-tree for the BASELINE geometry), and the work is cut differently:
//
//   * Stage A (tap sums).  A lane owns C CONSECUTIVE residues of one row (16 lanes x C = one row of q
//     residues; C = 11 for q = 169), a wave four rows.  A delta tap is C ds_read_b64 off ONE lane base with
//     immediate offsets (the ring slot of row m - a is a compile-time constant because the loop is unrolled
//     over the ring period: no per-tap address arithmetic, no table, no scalar loads), and a tooth that
//     enters or leaves as w adjacent taps is a SLIDING sum over C + w - 1 reads:
//     T_0 = e_0 + .. + e_{w-1},  T_i = T_{i-1} + e_{i+w-1} - e_{i-1}.  BASELINE geometry (28 delta taps =
//     14 single taps + two 7-wide teeth): 17.1 LDS reads and 20.7 float64 adds per output instead of 28 + 28
//     (+ 17 address adds + table traffic) in the phase-major kernel.
//   * Stage B (outputs) runs ONE ITERATION BEHIND stage A, on the first three waves, before they start their
//     (lighter) share of stage A.  Thread = residue (coalesced 8-byte stores as before) walks the NR rows of the
//     iteration: y = x - S / n_taps, S += Delta.  The Delta values change hands through a double-buffered LDS
//     array; S stays in a register for the whole stretch, so there is no chain between row groups at all; the
//     stores have a whole stage A to drain before the next barrier (one barrier per iteration).  (Stage B on
//     three waves of its own -- 11 waves -- caps the kernel at 168 VGPRs and hipcc then spills the prefetch
//     registers, whose loads are in flight: not an option.)
//   * Stage A runs on two wave sets (8 waves, one workgroup per CU: the ring of a second workgroup does not fit
//     the LDS): both sets work on the same NR rows, each on half of every lane's C outputs.  Two waves per SIMD
//     let one wave's LDS issue overlap the other's float64 adds (first version, one set of four waves: 5.6 ms of
//     pure issue time for 256 ch x 10 M).
//   * Rows are requested P = 4 iterations ahead into registers (16-byte coalesced buffer loads, ~90 KB in
//     flight per CU: the first version staged rows by LDS-DMA two iterations ahead, 48 KB in flight, and its
//     read stream alone ran at 4.1 TB/s) and copied into the ring one iteration ahead (ds_write_b128).  Ring
//     layout: row-major [slot][PITCH] with HB halo columns each side (loaded, not mirrored); PITCH = 4 mod 8 and
//     the two rows of a half-wave four slots apart make every ds_read_b64 conflict-free.
//   * Rows that touch the ends of the recording / of the addressable window are filled and emitted by a
//     generic slow path (zero padding, divisor from the cumulative tap counts), everything else is
//     straight-line code.
//
// Non-finite samples: as for every recurrence kernel, results are stored as computed and filter_repair_kernel
// runs behind the launch (parrm_filter_internal.h).
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "parrm_filter_internal.h"

namespace parrm_filter {

namespace {

constexpr int kNR = 16;       // rows per iteration (4 waves x 4 rows)
constexpr int kAThreads = 512;  // stage A: two wave sets of four waves (stage B: ceil(q / 64) more waves)
constexpr int kLdsLimit = 160 * 1024;
constexpr int kMaxGuard = 40;  // (what fits is decided by the LDS check below: 2 * guard more columns per ring row)

struct Unit {  // one delta element: `width` adjacent taps of one sign in row m - a, offsets b_lo .. b_lo + width - 1
    int a, b_lo, width, sign;
};
struct Run {   // full-tap run for the per-stretch initialisation
    int a, b_lo, b_hi;
};

struct Geom {
    int q = 0, c = 0, hb = 0, ncol = 0, pitch = 0, ms = 0, a_lo = 0, a_hi = 0;
    int pb = 0, dpb = 0, ring_bytes = 0, lds_total = 0, hs = 0, period = 0;
    int b_threads = 0, threads = 0;  // stage-B lanes (whole waves >= q); workgroup size
    int ofs = 0, ch16 = 0, d_bytes = 0, nld = 0;  // slot of row m: (m + ofs) mod ms; 16-byte chunks per row; one Delta buffer; loads per lane
    int n_taps = 0;
    int fsplit = 1;  // 2: rows of HALF the recurrence stride (strides above 176), two running sums per lane (row parity)
    std::vector<Unit> units;
    std::vector<Run> runs;
};

inline int floordiv_round(int64_t u, int64_t q) {  // nearest multiple
    return static_cast<int>(std::floor(static_cast<double>(u) / static_cast<double>(q) + 0.5));
}

// Geometry for a tap mask and a stride; false when this kernel cannot take the filter.
// `q`: the recurrence stride.  Up to 176 a ring row is one stride long.  Above (round 4: a 30 kHz recording of 130 Hz
// stimulation has T = 231) a row of q residues no longer fits 16 lanes x 11 and the LDS, so rows are HALF a stride
// (q even): sample n = m q/2 + rho, the delta taps u = a q/2 + b reach back an even number of such rows, and the
// running sum of a residue advances over TWO rows -- S(n + q) = S(n) + Delta(n) --, i.e. stage B keeps one sum for the
// even and one for the odd rows of its residue.  Everything else (ring, stage A, loader) is the kernel for stride q/2.
bool make_geom(const std::vector<int8_t> &tap, int64_t hw, int64_t q, Geom *g) {
    auto tap_at = [&](int64_t w) -> int { return (w >= -hw && w <= hw) ? tap[w + hw] : 0; };
    const int64_t stride = q;
    g->fsplit = 1;
    if (q > 176) {
        if (q & 1) return false;
        g->fsplit = 2;
        q /= 2;
    }
    // (16 lanes x C residues cover a row; the ring of 64+ rows and the two Delta buffers fit the LDS up to C = 11)
    if (q < 80 || q > 176) return false;
    g->q = static_cast<int>(q);
    int guard = 0;
    g->a_lo = 1 << 30;
    g->a_hi = -(1 << 30);
    // delta taps, ascending in u
    std::vector<std::pair<int64_t, int>> delta;
    for (int64_t u = -hw - stride; u <= hw; ++u) {
        const int d = tap_at(u + stride) - tap_at(u);
        if (d == 0) continue;
        delta.push_back({u, d});
        const int a = floordiv_round(u, q);
        const int b = static_cast<int>(u - static_cast<int64_t>(a) * q);
        guard = std::max(guard, std::abs(b));
        g->a_lo = std::min(g->a_lo, a);
        g->a_hi = std::max(g->a_hi, a);
    }
    if (delta.empty() || delta.size() > 256) return false;  // (96 until round 4: two whole 43-wide teeth at the comb's ends are 86 on their own)
    g->n_taps = 0;
    for (int64_t w = -hw; w <= hw; ++w) {
        if (!tap_at(w)) continue;
        ++g->n_taps;
        const int a = floordiv_round(w, q);
        guard = std::max(guard, std::abs(static_cast<int>(w - static_cast<int64_t>(a) * q)));
        g->a_lo = std::min(g->a_lo, a);  // the initialisation reads row -a of the prologue fill
        g->a_hi = std::max(g->a_hi, a);
    }
    // (the halo: 2 * guard more columns per ring row.  12 until round 4; wide teeth -- period_half_width of a tenth of
    // the period and more -- reach further, and as sliding sums they are cheap: C + w - 1 reads per tooth edge)
    if (guard > kMaxGuard) return false;
    g->hb = guard;
    // the ring always holds the rows of the current iteration themselves (stage B reads its own samples there), also
    // when every tap lies on one side of the centre (one-sided filters with omitted samples: found by the self-test)
    g->a_hi = std::max(g->a_hi, 0);
    g->a_lo = std::min(g->a_lo, 0);
    // units: maximal groups of adjacent offsets with one sign inside one row
    for (size_t i = 0; i < delta.size();) {
        const int a = floordiv_round(delta[i].first, q);
        size_t j = i;
        while (j + 1 < delta.size() && delta[j + 1].first == delta[j].first + 1 && delta[j + 1].second == delta[i].second &&
               floordiv_round(delta[j + 1].first, q) == a)
            ++j;
        Unit u;
        u.a = a;
        u.b_lo = static_cast<int>(delta[i].first - static_cast<int64_t>(a) * q);
        u.width = static_cast<int>(j - i + 1);
        u.sign = delta[i].second;
        g->units.push_back(u);
        i = j + 1;
    }
    for (int64_t w = -hw; w <= hw;) {
        if (!tap_at(w)) {
            ++w;
            continue;
        }
        const int a = floordiv_round(w, q);
        int64_t e = w;
        while (e + 1 <= hw && tap_at(e + 1) && floordiv_round(e + 1, q) == a) ++e;
        g->runs.push_back({a, static_cast<int>(w - static_cast<int64_t>(a) * q), static_cast<int>(e - static_cast<int64_t>(a) * q)});
        w = e + 1;
    }
    // C residues per lane, 16 lanes per row; C odd keeps a lane stride of C*8 bytes conflict-free
    int c = static_cast<int>((q + 15) / 16);
    if ((c & 1) == 0) ++c;
    if (c > 17) return false;
    g->c = c;
    g->ncol = 2 * g->hb + 16 * c;
    int pitch = g->ncol;
    while ((pitch & 7) != 4) ++pitch;  // 4 mod 8: partner rows four slots apart sit 32 banks apart; 16-byte rows
    g->pitch = pitch;
    g->pb = pitch * 8;
    g->dpb = 16 * c * 8;
    // Rows of iteration k+1 are copied into the ring during iteration k: live rows then are
    // [16k - max(a_hi, 16), 16(k+1) - a_lo + 16)  (stage B of iteration k-1 may still read its own rows).
    const int live = std::max(kNR, g->a_hi) + 2 * kNR - g->a_lo;
    g->ms = (live + kNR - 1) / kNR * kNR;
    g->period = g->ms / kNR;
    // slot of row m = (m + ofs) mod ms with ofs = a_lo (mod 16), so that the 16 rows loaded for one iteration
    // ([16k - a_lo, 16k - a_lo + 16)) never straddle the ring's end
    g->ofs = g->a_lo;
    while (g->ofs < g->a_hi) g->ofs += kNR;
    g->hs = g->ms / 2;
    g->ring_bytes = g->ms * g->pb;
    g->d_bytes = kNR * g->dpb;
    g->lds_total = g->ring_bytes + 2 * g->d_bytes;  // the Delta buffer is double-buffered
    g->b_threads = (g->q + 63) / 64 * 64;
    g->threads = kAThreads;  // stage B runs on the first b_threads / 64 waves, in front of their stage A
    g->ch16 = g->pb / 16;
    g->nld = (kNR * g->ch16 + kAThreads - 1) / kAThreads;
    if (g->lds_total > kLdsLimit) return false;
    if (g->hs * g->pb + g->pb > 65000) return false;         // immediates of ds_read are 16 bits
    if (g->ofs - g->a_lo + kNR > g->ms) return false;         // prologue fill must not wrap
    if (g->period < 2 || g->period > 6) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- source
struct Read {
    std::string base;
    int imm;
};

// (one name per pair of element types: a profiler's kernel trace tells the variants apart)
inline const char *kernel_name(bool in32, bool out32) {
    return !in32 ? "parrm_comb_kernel" : (out32 ? "parrm_comb_kernel_f32_f32" : "parrm_comb_kernel_f32_f64");
}

class Emitter {
   public:
    explicit Emitter(const Geom &g, int batch, bool in32 = false, bool out32 = false)
        : g_(g), batch_(batch), in32_(in32), out32_(out32) {
        // profiling ablations (results wrong by construction): 1 no tap reads, 2 no output stores, 4 no row requests,
        // 8 no second barrier, 16 no stage B at all, 32 no wait before the ring copy.  Part of the source text, so each has
        // its own code object.
        if (const char *e = getenv("PARRM_COMB_DEBUG")) debug_ = atoi(e);
        if (const char *e = getenv("PARRM_COMB_B16")) wide_b_ = atoi(e) != 0 && !in32 && !out32 && g.fsplit == 1;
        // cache policy of the once-read row requests (bit 0) and the once-written output stores (bit 1): `nt`
        // (part of the source text, like the ablations: every setting is a code object of its own)
        if (const char *e = getenv("PARRM_COMB_NT")) nt_ = atoi(e);
    }

    // (base register, immediate) of element `col` of ring row m - a for lane row r, at ring phase km
    Read tap_address(int km, int a, int col, std::vector<int> *thetas) const {
        int slot0 = (kNR * km + g_.ofs - a) % g_.ms;
        if (slot0 < 0) slot0 += g_.ms;
        Read r;
        if (slot0 + kNR - 1 < g_.ms) {
            if (slot0 < g_.hs) {
                r.base = "B0";
                r.imm = slot0 * g_.pb + col * 8;
            } else {
                r.base = "B1";
                r.imm = (slot0 - g_.hs) * g_.pb + col * 8;
            }
        } else {
            // lanes whose row wraps past the ring's end read RING_BYTES lower: one compare + select per unit (the
            // emitter declares the base in front of the unit's first read), not a register held for the whole kernel
            const int theta = g_.ms - slot0;
            if (std::find(thetas->begin(), thetas->end(), theta) == thetas->end()) thetas->push_back(theta);
            r.base = "B1w" + std::to_string(theta);
            r.imm = (slot0 - g_.hs) * g_.pb + col * 8;
        }
        return r;
    }

    // Pipelined LDS reads: statement j issues batch j and waits, with a counted lgkmcnt, for batch j - 1
    // (LDS operations return in order); the arithmetic on a batch is emitted behind the statement that waits
    // for it, fenced by a sched_barrier (the compiler does not know the results are asynchronous).
    //   reads[i]        : where value e<prefix><i> comes from
    //   compute[i]      : statements that may run once read i (and every earlier one) is back
    void emit_pipeline(std::ostringstream &o, const std::string &prefix, const std::vector<Read> &reads,
                       const std::vector<std::string> &compute, int batch = 0, bool wide = false) const {
        const int batch_ = batch ? batch : this->batch_;
        const int n = static_cast<int>(reads.size());
        const char *op = wide ? "ds_read_b128" : "ds_read_b64";
        o << (wide ? "        d2" : "        double");
        for (int i = 0; i < n; ++i) o << (i ? ", " : " ") << prefix << i;
        o << ";\n";
        const int nb = (n + batch_ - 1) / batch_;
        for (int j = 0; j <= nb; ++j) {
            if (j < nb) {
                const int lo = j * batch_, hi = std::min(n, lo + batch_);
                // distinct bases of this batch
                std::vector<std::string> bases;
                for (int i = lo; i < hi; ++i)
                    if (std::find(bases.begin(), bases.end(), reads[i].base) == bases.end()) bases.push_back(reads[i].base);
                o << "        asm volatile(";
                for (int i = lo; i < hi; ++i) {
                    const int bi = static_cast<int>(std::find(bases.begin(), bases.end(), reads[i].base) - bases.begin());
                    o << "\"" << op << " %" << (i - lo) << ", %" << (hi - lo + bi) << " offset:" << reads[i].imm << "\\n\\t\"\n                     ";
                }
                if (j > 0) o << "\"s_waitcnt lgkmcnt(" << (hi - lo) << ")\"";
                else o << "\"\"";
                o << "\n                     :";
                for (int i = lo; i < hi; ++i) o << (i > lo ? ", " : " ") << "\"=&v\"(" << prefix << i << ")";
                o << "\n                     :";
                for (size_t b = 0; b < bases.size(); ++b) o << (b ? ", " : " ") << "\"v\"(" << bases[b] << ")";
                o << ");\n";
            } else {
                o << "        asm volatile(\"s_waitcnt lgkmcnt(0)\");\n";
            }
            if (j > 0) {
                o << "        __builtin_amdgcn_sched_barrier(0);\n";
                const int lo = (j - 1) * batch_, hi = std::min(n, lo + batch_);
                for (int i = lo; i < hi; ++i)
                    if (!compute[i].empty()) o << compute[i];
                // (and nothing of it sinks below the next batch: the arithmetic would lag further and further behind
                // the reads and every value read would need a register of its own -- 56 VGPRs in stage B alone)
                o << "        __builtin_amdgcn_sched_barrier(0);\n";
            }
        }
    }

    // Stage A of ring phase km for outputs [i0, i1) of this lane's C: acc<i> = their Delta
    void emit_stage_a(std::ostringstream &o, int km, int i0, int i1) const {
        std::vector<int> local_thetas, *thetas = &local_thetas;
        std::vector<Read> reads;
        std::vector<std::string> compute;
        bool first = true;
        int tcount = 0;
        for (const Unit &u : g_.units) {
            const char *sg = u.sign > 0 ? "+" : "-";
            const int b_hi = u.b_lo + u.width - 1;
            const int base_col = g_.hb - b_hi;  // element 0 of the unit for output 0 (lane part l*C is in the base)
            const int first_read = static_cast<int>(reads.size());
            // elements j in [i0, i1 + width - 1): output i sums elements i .. i + width - 1
            const int n_el = (i1 - i0) + u.width - 1;
            for (int j = 0; j < n_el; ++j) {
                reads.push_back(tap_address(km, u.a, base_col + i0 + j, thetas));
                compute.push_back("");
            }
            auto e = [&](int j) { return "ea" + std::to_string(first_read + j); };  // j relative to i0
            if (u.width == 1) {
                for (int i = i0; i < i1; ++i) {
                    std::ostringstream s;
                    if (first) s << "        acc" << i << " = " << (u.sign > 0 ? "" : "-") << e(i - i0) << ";\n";
                    else s << "        acc" << i << " " << sg << "= " << e(i - i0) << ";\n";
                    compute[first_read + (i - i0)] = s.str();
                }
            } else {
                // sliding sum: T_i0 after element w-1, T_i after element (i - i0) + w - 1
                const std::string t = "tw" + std::to_string(tcount++);
                for (int i = i0; i < i1; ++i) {
                    std::ostringstream s;
                    const int ii = i - i0;
                    if (ii == 0 && u.width > 8) {
                        // a wide tooth (round 4): summed element by element as the reads come back -- the same
                        // left-to-right order -- so that only the C - 1 elements the sliding update subtracts
                        // later stay in registers, not all w of them (w = 20 ... 45: the build spilled)
                        compute[first_read] = "        double " + t + " = " + e(0) + ";\n";
                        for (int j = 1; j + 1 < u.width; ++j) compute[first_read + j] = "        " + t + " += " + e(j) + ";\n";
                        s << "        " << t << " += " << e(u.width - 1) << ";\n";
                    } else if (ii == 0) {
                        s << "        double " << t << " = " << e(0);
                        for (int j = 1; j < u.width; ++j) s << " + " << e(j);
                        s << ";\n";
                    } else {
                        s << "        " << t << " += " << e(ii + u.width - 1) << " - " << e(ii - 1) << ";\n";
                    }
                    if (first) s << "        acc" << i << " = " << (u.sign > 0 ? "" : "-") << t << ";\n";
                    else s << "        acc" << i << " " << sg << "= " << t << ";\n";
                    compute[first_read + ii + u.width - 1] = s.str();
                }
            }
            first = false;
        }
        if (debug_ & 1) {
            for (int i = i0; i < i1; ++i) o << "        acc" << i << " = 0.0;\n";
            return;
        }
        // (r goes through an empty asm so that these are NOT loop-invariant to the compiler: hoisted out of the
        // loop they would hold a dozen registers for the whole kernel)
        if (!local_thetas.empty()) o << "        int r_now = r;\n        asm volatile(\"\" : \"+v\"(r_now));\n";
        for (int th : local_thetas) o << "        const u32 B1w" << th << " = B1 - (r_now >= " << th << " ? (u32)RING_BYTES : 0u);\n";
        emit_pipeline(o, "ea", reads, compute);
    }

    // Stage B of ring phase km, straight-line form (all NR rows interior): tid < Q
    void emit_stage_b(std::ostringstream &o, int km) const {
        std::vector<Read> reads;
        std::vector<std::string> compute;
        for (int r = 0; r < kNR; ++r) {
            const int slot = (kNR * km + g_.ofs + r) % g_.ms;
            Read x;
            if (slot < g_.hs) {
                x.base = "xb0";
                x.imm = slot * g_.pb;
            } else {
                x.base = "xb1";
                x.imm = (slot - g_.hs) * g_.pb;
            }
            reads.push_back(x);
            compute.push_back("");
            reads.push_back({"dbk", r * g_.dpb});
            std::ostringstream s;
            const int ix = 2 * r;
            const char *sum = (g_.fsplit == 2 && (r & 1)) ? "SO" : "S";  // (NR is even: the parity of a row is that of r)
            s << "        { const double yv = __builtin_fma(-" << sum << ", inv_taps, eb" << ix << ");\n";
            if (debug_ & 2) s << "          asm volatile(\"\" :: \"v\"(yv));\n";
            else if (out32_) s << "          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, (float)yv), rsrc_y, yoff, " << r * g_.q * 4 << "u, " << ((nt_ & 2) ? 2 : 0) << ");\n";
            else s << "          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yv), rsrc_y, yoff, " << r * g_.q * 8 << "u, " << ((nt_ & 2) ? 2 : 0) << ");\n";
            s << "          " << sum << " += eb" << ix + 1 << "; }\n";
            compute.push_back(s.str());
        }
        emit_pipeline(o, "eb", reads, compute, 6);  // three rows per batch
    }

    // The same with TWO residues per lane (2j, 2j + 1): 16-byte LDS reads and 16-byte output stores on
    // ceil(Q / 2) lanes; an odd Q leaves the last lane one residue, stored by an 8-byte store of its own.
    void emit_stage_b_wide(std::ostringstream &o, int km) const {
        std::vector<Read> reads;
        std::vector<std::string> compute;
        for (int r = 0; r < kNR; ++r) {
            const int slot = (kNR * km + g_.ofs + r) % g_.ms;
            Read x;
            if (slot < g_.hs) {
                x.base = "xb0";
                x.imm = slot * g_.pb;
            } else {
                x.base = "xb1";
                x.imm = (slot - g_.hs) * g_.pb;
            }
            reads.push_back(x);
            compute.push_back("");
            reads.push_back({"dbk", r * g_.dpb});
            std::ostringstream s;
            const int ix = 2 * r;
            s << "        { d2 yv; yv.x = __builtin_fma(-S, inv_taps, eb" << ix << ".x); yv.y = __builtin_fma(-S1, inv_taps, eb" << ix << ".y);\n";
            if (debug_ & 2) {
                s << "          asm volatile(\"\" :: \"v\"(yv));\n";
            } else if (g_.q & 1) {
                s << "          if (rho2 + 1 < Q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), rsrc_y, yoff, " << r * g_.q * 8 << "u, 0);\n"
                  << "          else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yv.x), rsrc_y, yoff, " << r * g_.q * 8 << "u, 0);\n";
            } else {
                s << "          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), rsrc_y, yoff, " << r * g_.q * 8 << "u, 0);\n";
            }
            s << "          S += eb" << ix + 1 << ".x; S1 += eb" << ix + 1 << ".y; }\n";
            compute.push_back(s.str());
        }
        emit_pipeline(o, "eb", reads, compute, 6, true);
    }

    // Request the NR rows of iteration `kk` (an expression) into register set `set`.  The loads are inline asm and
    // the wait before the set is copied into the ring is a COUNTED vmcnt written by hand (emit_write_iter): with
    // compiler-visible loads hipcc drained the queue (vmcnt(0)) once per ring period, i.e. gave up the whole
    // prefetch depth every P iterations.  Loads retire in issue order among themselves, so "at most N
    // outstanding" with N = the loads issued after the set's own on the regular path (those of the P-1 iterations
    // since) means the set has landed; a path that issues fewer (rows at the ends of the window are fetched by
    // compiler-visible loads) raises `irregular` and the next P waits drain.
    void emit_load_iter(std::ostringstream &o, int set, const std::string &kk) const {
        if (debug_ & 4) {
            for (int i = 0; i < g_.nld; ++i) o << "        pf" << set << "_" << i << " = PFZERO;\n";
            return;
        }
        // (nothing is requested beyond the stretch's last iteration: a load that lands after the loop would write
        // into registers the compiler has reused by then -- it did, and corrupted the running sums of the last
        // rows of a few stretches per launch; `irregular` makes the waits of those last iterations drain)
        o << "        if ((" << kk << ") >= n_iter) {\n            irregular = " << g_.period << ";\n        } else {\n"
          << "        const long long lo = s0 + (long long)(NR * (" << kk << ") - ALO) * Q - HB;\n"
          << "        if (lo >= lim_lo && lo + (NR - 1) * Q + NCOL <= lim_hi) {\n"
          << "            const u32 soff = (u32)((lo - a.buf_first) * XB);\n            asm volatile(";
        const int n = g_.nld;
        // (float32 rows: 8-byte requests at 4-byte aligned addresses -- fine for buffer loads, checked on gfx950)
        for (int i = 0; i < n; ++i)
            o << "\"" << (in32_ ? "buffer_load_dwordx2" : "buffer_load_dwordx4") << " %" << i << ", %" << n + i << ", %" << 2 * n << ", %" << 2 * n + 1 << " offen" << ((nt_ & 1) ? " nt" : "") << "\\n\\t\"\n                         ";
        o << "\"\"\n                         :";
        for (int i = 0; i < n; ++i) o << (i ? ", " : " ") << "\"=&v\"(pf" << set << "_" << i << ")";
        o << "\n                         :";
        for (int i = 0; i < n; ++i) o << (i ? ", " : " ") << "\"v\"(goff" << i << ")";
        o << ", \"s\"(xdesc), \"s\"(soff) : \"memory\");\n";
        o << "        } else {\n            irregular = " << g_.period << ";\n";
        for (int i = 0; i < g_.nld; ++i)
            o << "            pf" << set << "_" << i << " = load_pair_padded(xrow, a.buf_first, lim_lo, lim_hi, (goff" << i << " >> 31) ? -(1ll << 40) : lo + (long long)(goff" << i << " >> XSH));\n";
        // (the empty asm makes the compiler wait for these loads HERE: left pending they would put its own
        // vmcnt(0) in front of the ring copy of every iteration that merges with this path)
        for (int i = 0; i < g_.nld; ++i) o << "            asm volatile(\"\" : \"+v\"(pf" << set << "_" << i << "));\n";
        o << "        }\n        }\n";
    }

    // copy register set `set` into the ring slots that start at byte offset `sb`
    void emit_write_iter(std::ostringstream &o, int set, int sb) const {
        const bool partial = kNR * g_.ch16 % kAThreads != 0;
        const int younger_loads = (g_.period - 1) * g_.nld;
        // Only LOADS are counted, on the stage-B waves too: their output stores sit between the loads in issue
        // order, but a count that includes them (57 = 9 loads + 48 stores) is satisfied as soon as the stores have
        // retired, whatever the loads are doing -- stores retire ahead of older loads often enough to corrupt a few
        // stretches per launch (measured).  "At most 9 outstanding" can only be the 9 youngest loads, because loads
        // retire in order among themselves; it merely makes those waves wait for stores issued a whole stage A ago.
        if (!(debug_ & 32))
            o << "        if (irregular > 0) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); --irregular; }\n"
              << "        else asm volatile(\"s_waitcnt vmcnt(" << std::min(63, younger_loads) << ")\" ::: \"memory\");\n";
        // (a float32 recording is widened by ordinary arithmetic, which -- unlike the LDS store -- the "memory" clobber
        // of the wait does not hold back: the ties make the conversions depend on the wait)
        if (in32_)
            for (int i = 0; i < g_.nld; ++i) o << "        asm volatile(\"\" : \"+v\"(pf" << set << "_" << i << "));\n";
        for (int i = 0; i < g_.nld; ++i) {
            o << "        ";
            if (partial && i == g_.nld - 1) o << "if (lvalid) ";
            o << "*(LDS_AS u32x4 *)(loff" << i << " + " << sb << "u) = widen(pf" << set << "_" << i << ");\n";
        }
    }

    std::string source() const {
        std::ostringstream o;
        const Geom &g = g_;
        const int P = g.period;
        o << "// generated by parrm_filter_comb.hip -- q " << g.q << ", C " << g.c << ", " << g.units.size() << " delta units\n";
        if (g.fsplit == 2) o << "// rows of HALF the recurrence stride (" << 2 * g.q << "): running sums S (even rows) and SO (odd rows)\n";
        o << "typedef unsigned int u32;\ntypedef unsigned int u32x2 __attribute__((ext_vector_type(2)));\n"
          << "typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));\ntypedef double d2 __attribute__((ext_vector_type(2)));\n";
        // element types: XT recording (XB bytes, XSH = log2), YT output (YB bytes); PFT = what one row request returns
        // per lane (two samples) -- a float32 recording is widened to float64 on its way into the ring (widen()), the
        // arithmetic is float64 throughout, a float32 output is rounded once at the store
        o << "typedef " << (in32_ ? "float" : "double") << " XT;\n#define XB " << (in32_ ? 4 : 8) << "\n#define XSH " << (in32_ ? 2 : 3)
          << "\ntypedef " << (out32_ ? "float" : "double") << " YT;\n#define YB " << (out32_ ? 4 : 8) << "\ntypedef "
          << (in32_ ? "u32x2" : "u32x4") << " PFT;\n#define PFZERO " << (in32_ ? "(u32x2{0u, 0u})" : "(u32x4{0u, 0u, 0u, 0u})")
          << "\n#define KNAME " << kernel_name(in32_, out32_) << "\n";
        o << "#define LDS_AS __attribute__((address_space(3)))\n";
        o << "#define Q " << g.q << "\n#define CC " << g.c << "\n#define NR " << kNR << "\n#define MS " << g.ms << "\n#define HB " << g.hb
          << "\n#define NCOL " << g.ncol << "\n#define PB " << g.pb << "\n#define DPB " << g.dpb << "\n#define AHI " << g.a_hi
          << "\n#define ALO (" << g.a_lo << ")\n#define OFS " << g.ofs << "\n#define RING_BYTES " << g.ring_bytes << "\n#define DBYTES " << g.d_bytes
          << "\n#define LDS_TOTAL " << g.lds_total << "\n#define HSTEP " << g.hs * g.pb << "\n#define CH16 " << g.ch16
          << "\n#define PERIOD " << g.period << "\n#define ATHREADS " << kAThreads << "\n#define BTHREADS " << (wide_b_ ? ((g.q + 1) / 2 + 63) / 64 * 64 : g.b_threads) << "\n#define RPL " << (wide_b_ ? 2 : 1) << "\n#define NTHREADS " << g.threads << "\n#define N_RUNS " << g.runs.size() << "\n";
        o << "__device__ const int RUN_A[N_RUNS] = {";
        for (size_t i = 0; i < g.runs.size(); ++i) o << (i ? "," : "") << g.runs[i].a;
        o << "};\n__device__ const int RUN_BLO[N_RUNS] = {";
        for (size_t i = 0; i < g.runs.size(); ++i) o << (i ? "," : "") << g.runs[i].b_lo;
        o << "};\n__device__ const int RUN_BHI[N_RUNS] = {";
        for (size_t i = 0; i < g.runs.size(); ++i) o << (i ? "," : "") << g.runs[i].b_hi;
        o << "};\n";
        // running tap sum of row 0, tap by tap (thread = residue): generated straight-line with the ring offsets as
        // constants and four partial sums, so that the reads are in flight together (a loop over the run table
        // was a chain of ~200 dependent LDS round trips, ~7 us per stretch)
        std::ostringstream sd;
        sd << "    double S = 0.0, S1 = 0.0;\n    if (rho < Q) {\n        const double *sdp = reinterpret_cast<const double *>(smem) + (HB + rho);\n"
           << "        double sa0 = 0.0, sa1 = 0.0, sa2 = 0.0, sa3 = 0.0, sb0 = 0.0, sb1 = 0.0, sb2 = 0.0, sb3 = 0.0;\n";
        {
            int n = 0;
            for (const Run &r : g.runs)
                for (int b = r.b_lo; b <= r.b_hi; ++b, ++n) {
                    const int el = (g.ofs - r.a) * g.pitch - b;
                    sd << "        sa" << (n & 3) << " += sdp[" << el << "];";
                    if (wide_b_) sd << " sb" << (n & 3) << " += sdp[" << el + 1 << "];";
                    sd << "\n";
                }
        }
        sd << "        S = (sa0 + sa1) + (sa2 + sa3);\n        S1 = (sb0 + sb1) + (sb2 + sb3);\n    }\n";
        if (g.fsplit == 2) {  // the same for row 1: the running sum of the odd rows
            sd << "    double SO = 0.0;\n    if (rho < Q) {\n        const double *sdp = reinterpret_cast<const double *>(smem) + (HB + rho);\n"
               << "        double so0 = 0.0, so1 = 0.0, so2 = 0.0, so3 = 0.0;\n";
            int n = 0;
            for (const Run &r : g.runs)
                for (int b = r.b_lo; b <= r.b_hi; ++b, ++n)
                    sd << "        so" << (n & 3) << " += sdp[" << (g.ofs + 1 - r.a) * g.pitch - b << "];\n";
            sd << "        SO = (so0 + so1) + (so2 + so3);\n    }\n";
        }
        const std::string sdirect = sd.str();
        o << R"SRC(
struct CombArgs {
    const XT *x;
    YT *y;
    long long n_chans, buf_first, buf_len, out_first, out_len, n_total, ldx, ldy;
    const int *tapcum;
    long long stretch_len, n_stretch;
    double inv_taps;
    int hw, pad;
};

__device__ __forceinline__ u32 lds_off(const void *p) { return (u32)(size_t)(const LDS_AS void *)p; }

// one output of a row that touches an end of the recording (divisor = taps inside the recording, parrm.py:862-866)
__device__ __noinline__ void emit_edge(YT *yout, const int *tapcum, long long hw, long long n_total, double inv_taps,
                                       long long n, double xc, double s) {
    double y;
    if (n >= hw && n + hw < n_total) {
        y = __builtin_fma(-s, inv_taps, xc);
    } else {
        const long long w_hi = n < hw ? n : hw;
        long long w_lo = n - n_total + 1;
        if (w_lo < -hw) w_lo = -hw;
        int v = 0;
        if (w_hi >= w_lo) v = tapcum[w_hi + hw + 1] - tapcum[w_lo + hw];
        y = v > 0 ? xc - s / (double)v : s * 0.0;  // no tap inside: 0, a poisoned sum stays visible for the repair pass
    }
    *yout = (YT)y;
}

// two samples as the ring holds them (float64)
__device__ __forceinline__ u32x4 widen(PFT v) {
#if XB == 8
    return v;
#else
    // (the components through scalars of their own: __builtin_bit_cast(float, v.y) on a vector component compiles to
    // component 0 with this hipcc -- both halves of the chunk came out as the first sample)
    const u32 s0_ = v.x, s1_ = v.y;
    const u32x2 w0 = __builtin_bit_cast(u32x2, (double)__builtin_bit_cast(float, s0_));
    const u32x2 w1 = __builtin_bit_cast(u32x2, (double)__builtin_bit_cast(float, s1_));
    return u32x4{w0.x, w0.y, w1.x, w1.y};
#endif
}

// samples n, n + 1 of a row, zero outside [lim_lo, lim_hi) (the ends of the recording / of the addressable window)
__device__ __noinline__ PFT load_pair_padded(const XT *xrow, long long buf_first, long long lim_lo, long long lim_hi, long long n) {
    const XT v0 = (n >= lim_lo && n < lim_hi) ? xrow[n - buf_first] : (XT)0;
    const XT v1 = (n + 1 >= lim_lo && n + 1 < lim_hi) ? xrow[n + 1 - buf_first] : (XT)0;
#if XB == 8
    const u32x2 w0 = __builtin_bit_cast(u32x2, v0), w1 = __builtin_bit_cast(u32x2, v1);
    return u32x4{w0.x, w0.y, w1.x, w1.y};
#else
    return u32x2{__builtin_bit_cast(u32, v0), __builtin_bit_cast(u32, v1)};
#endif
}

extern "C" __global__ void __launch_bounds__(NTHREADS) KNAME(CombArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
    const u32 ring = lds_off(smem);
    const u32 dbuf = ring + RING_BYTES;
    const int tid = threadIdx.x;
    // (explicitly wave-uniform: the buffer descriptors and row offsets below must live in scalar registers)
    const u32 n_str = (u32)a.n_stretch;
    const u32 ch32 = __builtin_amdgcn_readfirstlane(blockIdx.x / n_str);
    const long long ch = ch32;
    const long long st = blockIdx.x - ch32 * n_str;
    const long long s0 = a.out_first + st * a.stretch_len;  // sample of (row 0, residue 0)
    long long s_end = s0 + a.stretch_len;
    if (s_end > a.out_first + a.out_len) s_end = a.out_first + a.out_len;
    const int rows_total = (int)((s_end - s0 + Q - 1) / Q);
    const int n_iter = (rows_total + NR - 1) / NR;
    const XT *xrow = a.x + ch * a.ldx;  // sample n at xrow[n - buf_first]
    const long long lim_lo = a.buf_first > 0 ? a.buf_first : 0;
    long long lim_hi = a.buf_first + a.buf_len;
    if (lim_hi > a.n_total) lim_hi = a.n_total;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT *>(xrow), 0, (int)(a.buf_len * XB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y + ch * a.ldy, 0, (int)(a.out_len * YB), 0x00020000);
    const double inv_taps = a.inv_taps;
    // the same descriptor as four scalars, for the inline-asm row loads
    u32x4 xdesc;
    {
        const unsigned long long xp = (unsigned long long)xrow;
        xdesc.x = __builtin_amdgcn_readfirstlane((u32)xp);
        xdesc.y = __builtin_amdgcn_readfirstlane((u32)(xp >> 32) & 0xffffu);
        xdesc.z = __builtin_amdgcn_readfirstlane((u32)(a.buf_len * XB));
        xdesc.w = 0x00020000u;
    }
    // > 0: the counted vmcnt of the ring copy does not hold (rows at the ends of the window are fetched by a
    // different number of operations than the regular path issues) -> that wait drains the queue
    int irregular = 0;
    // every wave computes tap sums (stage A); the first ceil(Q / 64) waves also turn the tap sums of the PREVIOUS
    // iteration into outputs (stage B) before they start on this one
    const bool wave_has_b = __builtin_amdgcn_readfirstlane(tid) < BTHREADS;
    const int rho = tid * RPL;  // stage-B lane: its first residue (RPL residues per lane)

    // prologue: rows [-AHI, -ALO + NR) -> slots OFS - AHI .. (everything iteration 0 reads), 16-byte chunks
    {
        const long long p_lo = s0 - (long long)AHI * Q - HB;  // first sample of the fill
        if (p_lo >= lim_lo && p_lo + (long long)(AHI - ALO + NR - 1) * Q + NCOL <= lim_hi) {
            // every row is addressable (all stretches but the first and the last of a recording): 16-byte buffer
            // loads, all in flight together
            const u32 soff = (u32)((p_lo - a.buf_first) * XB);
#pragma unroll 3
            for (int t = tid; t < (AHI - ALO + NR) * CH16; t += NTHREADS) {
                const int row = t / CH16, c16 = t - row * CH16;
#if XB == 8
                const PFT v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (u32)(row * (Q * 8) + c16 * 16), soff, 0);
#else
                const PFT v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, (u32)(row * (Q * 4) + c16 * 8), soff, 0);
#endif
                *(LDS_AS u32x4 *)(ring + (u32)((row + OFS - AHI) * PB + c16 * 16)) = widen(v);
            }
        } else {
#pragma unroll 1
            for (int t = tid; t < (AHI - ALO + NR) * CH16; t += NTHREADS) {
                const int row = t / CH16, c16 = t - row * CH16;
                const long long n = p_lo + (long long)row * Q + 2 * c16;
                *(LDS_AS u32x4 *)(ring + (u32)((row + OFS - AHI) * PB + c16 * 16)) = widen(load_pair_padded(xrow, a.buf_first, lim_lo, lim_hi, n));
            }
        }
    }
    __syncthreads();
    // running tap sum of row 0, evaluated tap by tap (thread = residue)
)SRC";
        o << sdirect;
        o << R"SRC(    // stage-A lane: wave set hs takes its half of the delta units for row r of the iteration, residues [l*CC, l*CC + CC)
    const int hs = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int gi = (tid & 255) >> 4, l = tid & 15;
    const int r = (gi >> 2) + 4 * (gi & 3);
    const u32 B0 = ring + r * PB + l * (CC * 8);
    const u32 B1 = B0 + HSTEP;
)SRC";
        // per-lane slots of the row loads: chunk t = i * NTHREADS + tid of the NR x CH16 chunks of an iteration
        for (int i = 0; i < g.nld; ++i) {
            o << "    const int lt" << i << " = " << i << " * ATHREADS + tid, lrow" << i << " = lt" << i << " / CH16, lc" << i << " = lt" << i
              << " - lrow" << i << " * CH16;\n";
            const bool last_partial = (i == g.nld - 1) && (kNR * g.ch16 % kAThreads != 0);
            if (last_partial) {
                o << "    const bool lvalid = lt" << i << " < NR * CH16;\n";
                o << "    const u32 goff" << i << " = lvalid ? (u32)(lrow" << i << " * (Q * XB) + lc" << i << " * (2 * XB)) : 0x80000000u;\n";

            } else {
                o << "    const u32 goff" << i << " = (u32)(lrow" << i << " * (Q * XB) + lc" << i << " * (2 * XB));\n";

            }
            o << "    const u32 loff" << i << " = ring + (u32)(lrow" << i << " * PB + lc" << i << " * 16);\n";
        }
        for (int set = 0; set < P; ++set)
            for (int i = 0; i < g.nld; ++i) o << "    PFT pf" << set << "_" << i << " = PFZERO;\n";

        // the two stage-A wave sets split each lane's C outputs; set 0 holds the stage-B waves and takes fewer
        int c_split = (g.c * 3 + 5) / 11;  // 3 of 11 (measured: 2 and 3 tie, 4 is 1 % slower, 5 is 2.5 % slower)
        if (const char *e = getenv("PARRM_COMB_SPLIT")) c_split = atoi(e);
        c_split = std::min(g.c - 1, std::max(1, c_split));
        std::ostringstream bodies;
        for (int km = 0; km < P; ++km) {
            const int kmb = (km + P - 1) % P;  // ring phase of iteration k - 1
            bodies << "        // ------------------------------------------------------------ ring phase " << km << "\n        {\n";
            // 0. stage B of iteration k - 1 (its stores then have the whole of stage A to drain)
            bodies << "        if (wave_has_b && k > 0) {\n        const int kk = k - 1;\n"
                   << "        const long long nrow = s0 + (long long)kk * (NR * Q);\n"
                   << "        const bool fast = (kk + 1) * NR * (long long)Q + s0 <= s_end && nrow >= a.hw && nrow + NR * Q - 1 + a.hw < a.n_total;\n"
                   << "        if (fast) {\n          if (rho < Q" << ((debug_ & 16) ? " && false" : "") << ") {\n"
                   << "          const u32 yoff = ystart + (u32)kk * (u32)(NR * Q * YB);\n          const u32 dbk = db + (u32)(kk & 1) * DBYTES;\n"
                   << "          const int rho2 = rho;\n          (void)rho2;\n";
            if (wide_b_) emit_stage_b_wide(bodies, kmb);
            else emit_stage_b(bodies, kmb);
            bodies << "          }\n        } else {\n          stage_b_generic(kk);\n        }\n        }\n";
            // 1. request the rows of iteration k + P
            emit_load_iter(bodies, km, "k + " + std::to_string(P));
            // 2. stage A: this wave set's part of the lane's outputs
            bodies << "        double";
            for (int i = 0; i < g.c; ++i) bodies << (i ? ", " : " ") << "acc" << i;
            bodies << ";\n        if (hs == 0) {\n";
            emit_stage_a(bodies, km, 0, c_split);
            bodies << "        } else {\n";
            emit_stage_a(bodies, km, c_split, g.c);
            bodies << "        }\n";
            // 3. Delta values into buffer k & 1 (stage B of iteration k - 1 read the other one)
            bodies << "        const u32 dw = dwr + (u32)(k & 1) * DBYTES;\n        if (hs == 0) {\n";
            for (int i = 0; i < c_split; ++i)
                bodies << "            *(LDS_AS double *)(dw + " << 8 * i << "u) = acc" << i << ";\n";
            bodies << "        } else {\n";
            for (int i = c_split; i < g.c; ++i)
                bodies << "            *(LDS_AS double *)(dw + " << 8 * i << "u) = acc" << i << ";\n";
            bodies << "        }\n";
            // 4. rows of iteration k + 1 (requested P - 1 iterations ago) into the ring
            const int sb1 = ((kNR * ((km + 1) % P) + g.ofs - g.a_lo) % g.ms) * g.pb;
            emit_write_iter(bodies, (km + 1) % P, sb1);
            bodies << "        asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n        }\n";
            bodies << "        if (++k >= n_iter) break;\n";
        }
        o << R"SRC(    const u32 dwr = dbuf + r * DPB + l * (CC * 8);
    // stage-B lane
    const u32 xb0 = ring + (HB + rho) * 8, xb1 = xb0 + HSTEP;
    const u32 db = dbuf + rho * 8;
    const u32 ystart = (u32)((s0 - a.out_first) * YB) + (u32)rho * (u32)YB;
    // stage B of iteration kk for rows that touch an end of the recording or of the stretch (and the last iteration)
    auto stage_b_generic = [&](int kk) {
        if (rho < Q) {
#pragma unroll 1
            for (int rr = 0; rr < NR; ++rr) {
                const int sl = (NR * kk + OFS + rr) % MS;
                const double *xp = reinterpret_cast<const double *>(smem + sl * PB + (HB + rho) * 8);
                const double *dp = reinterpret_cast<const double *>(smem + RING_BYTES + (kk & 1) * DBYTES + rr * DPB + rho * 8);
                const int m = kk * NR + rr;
                const long long n = s0 + (long long)m * Q + rho;
                if (m < rows_total && n < s_end) emit_edge(a.y + ch * a.ldy + (n - a.out_first), a.tapcum, a.hw, a.n_total, inv_taps, n, xp[0], S);
                S += dp[0];
                if (RPL == 2 && rho + 1 < Q) {
                    if (m < rows_total && n + 1 < s_end) emit_edge(a.y + ch * a.ldy + (n + 1 - a.out_first), a.tapcum, a.hw, a.n_total, inv_taps, n + 1, xp[1], S1);
                    S1 += dp[1];
                }
            }
        }
    };
)SRC";
        // rows of iterations 1 .. P-1 are on their way before the loop starts
        for (int set = 1; set < P; ++set) emit_load_iter(o, set, std::to_string(set));
        o << "    int k = 0;\n    while (true) {\n";
        o << bodies.str();
        o << "    }\n    if (wave_has_b) stage_b_generic(n_iter - 1);\n}\n";
        std::string text = o.str();
        if (g.fsplit == 2) {  // the generic (edge) form of stage B: the row's parity picks the running sum
            auto replace_all = [&](const std::string &from, const std::string &to) {
                size_t at = 0, n = 0;
                while ((at = text.find(from, at)) != std::string::npos) {
                    text.replace(at, from.size(), to);
                    at += to.size();
                    ++n;
                }
                return n;
            };
            const size_t a = replace_all("inv_taps, n, xp[0], S);", "inv_taps, n, xp[0], ((m & 1) ? SO : S));");
            const size_t b = replace_all("                S += dp[0];", "                ((m & 1) ? SO : S) += dp[0];");
            if (a != 1 || b != 1) return std::string();  // (the template text changed under this patch: no kernel)
        }
        return text;
    }

   private:
    const Geom &g_;
    int batch_;
    int debug_ = 0;
    int nt_ = 0;
    bool wide_b_ = false;  // stage B: two residues per lane, 16-byte reads and stores
    bool in32_ = false, out32_ = false;  // float32 recording (widened to float64 on its way into the ring) / float32 output
};

// ---------------------------------------------------------------------------------------------- code objects
uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) {
        h ^= ch;
        h *= 1099511628211ull;
    }
    return h;
}

std::string library_dir() {
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&fnv1a), &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        if (k != std::string::npos) return p.substr(0, k);
    }
    return ".";
}

std::string user_cache_dir() {
    if (const char *e = getenv("PARRM_KERNEL_CACHE")) return e;
    if (const char *h = getenv("HOME")) return std::string(h) + "/.cache/pyparrm_amd";
    // (no HOME: a directory of this user's own under /tmp -- never one another user can have made first)
    return "/tmp/pyparrm_amd_cache_" + std::to_string(static_cast<long long>(getuid()));
}

// A cache directory's code objects are EXECUTED: it is read only when it belongs to this user and nobody else can
// write to it (the round-3 fallback /tmp/pyparrm_amd_cache was world-visible, created 0755 by whoever got
// there first; another local user could have planted comb_<hash>.hsaco).  The in-tree <library dir>/kernels/ is part
// of the installation and trusted like the library itself.
bool cache_dir_is_safe(const std::string &dir) {
    struct stat sb;
    if (stat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) return false;
    return sb.st_uid == getuid() && (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

bool read_file(const std::string &path, std::vector<char> *out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return !out->empty();
}

bool write_file_atomic(const std::string &path, const std::vector<char> &data) {
    const std::string tmp = path + ".tmp" + std::to_string(static_cast<long long>(getpid()));
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return false;
        f.write(data.data(), static_cast<std::streamsize>(data.size()));
        if (!f) return false;
    }
    return rename(tmp.c_str(), path.c_str()) == 0;
}

void make_dirs(const std::string &dir) {
    std::string acc;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!acc.empty()) (void)mkdir(acc.c_str(), i == dir.size() ? 0700 : 0755);  // (the cache directory itself: this user only)
        }
        if (i < dir.size()) acc.push_back(dir[i]);
    }
}

bool compile_source(const std::string &src, std::vector<char> *code, std::string *log) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "parrm_comb.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        *log = "hiprtcCreateProgram failed";
        return false;
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
    size_t ls = 0;
    if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        log->resize(ls);
        (void)hiprtcGetProgramLog(prog, &(*log)[0]);
    }
    bool ok = rc == HIPRTC_SUCCESS;
    if (ok) {
        size_t cs = 0;
        ok = hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (ok) {
            code->resize(cs);
            ok = hiprtcGetCode(prog, code->data()) == HIPRTC_SUCCESS;
        }
    }
    (void)hiprtcDestroyProgram(&prog);
    return ok;
}

// The name carries a hash of the source AND of what turns it into code: the target and the hipRTC version (a code
// object built by another compiler release is not looked up).
std::string code_name(const std::string &src) {
    // (asked ONCE: every hipRTC entry point takes the library's global lock, which hiprtcCompileProgram holds for the
    // whole compile -- a version query from the launching thread sat out a background build's remaining 250-330 ms)
    static const std::string salt = [] {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        char buf[64];
        snprintf(buf, sizeof buf, "|gfx950|hiprtc %d.%d", major, minor);
        return std::string(buf);
    }();
    char buf[64];
    snprintf(buf, sizeof buf, "comb_%016llx.hsaco", static_cast<unsigned long long>(fnv1a(src + salt)));
    return buf;
}

// code object for a source text: in-tree directory, user cache, else hipRTC (result stored in the user cache)
bool code_for_source(const std::string &src, std::vector<char> *code, std::string *why) {
    const std::string name = code_name(src);
    if (!getenv("PARRM_COMB_NO_CACHE")) {
        if (read_file(library_dir() + "/kernels/" + name, code)) return true;
        if (cache_dir_is_safe(user_cache_dir()) && read_file(user_cache_dir() + "/" + name, code)) return true;
    }
    std::string log;
    if (!compile_source(src, code, &log)) {
        *why = "hipRTC: " + log;
        return false;
    }
    make_dirs(user_cache_dir());
    if (cache_dir_is_safe(user_cache_dir())) (void)write_file_atomic(user_cache_dir() + "/" + name, *code);
    return true;
}

int pick_batch() {
    int b = 7;
    if (const char *e = getenv("PARRM_COMB_BATCH")) b = atoi(e);
    return std::min(7, std::max(2, b));
}

// device struct of the generated source
struct CombArgs {
    const void *x;  // (XT * / YT * in the generated source)
    void *y;
    long long n_chans, buf_first, buf_len, out_first, out_len, n_total, ldx, ldy;
    const int *tapcum;
    long long stretch_len, n_stretch;
    double inv_taps;
    int hw, pad;
};

}  // namespace

struct CombKernel {
    Geom geom;
    bool in32 = false, out32 = false;  // element types of the recording / the output (float32, else float64)
    std::string source;
    hipModule_t module = nullptr;
    hipFunction_t func = nullptr;
    std::string error;
};

// Geometry + source for a tap mask (no device needed); nullptr when this kernel cannot take the filter.
// `attempt` > 0: leaner variants for geometries whose first build spills registers (smaller read batches keep
// fewer values in flight): the caller walks 0, 1, 2 until a build has no scratch.
CombKernel *comb_generate(const std::vector<int8_t> &tap, int64_t hw, int64_t q, int attempt, bool in32, bool out32) {
    Geom g;
    if (!make_geom(tap, hw, q, &g)) return nullptr;
    if (attempt > 4) return nullptr;
    auto *k = new CombKernel();
    k->geom = g;
    k->in32 = in32;
    k->out32 = out32;
    const int batch = attempt == 0 ? pick_batch() : attempt == 1 ? 5 : attempt == 2 ? 4 : attempt == 3 ? 3 : 2;
    k->source = Emitter(k->geom, batch, in32, out32).source();
    if (k->source.empty()) {
        delete k;
        return nullptr;
    }
    return k;
}

void comb_destroy(CombKernel *k) {
    if (!k) return;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
}

// Loads (compiling if no cached code object exists) the kernel on the current device.
bool comb_load(CombKernel *k) {
    if (k->func) return true;
    std::vector<char> code;
    if (!code_for_source(k->source, &code, &k->error)) return false;
    hipError_t e = hipModuleLoadData(&k->module, code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&k->func, k->module, kernel_name(k->in32, k->out32));
    int scratch = 0;
    if (e == hipSuccess) e = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->func);
    if (e == hipSuccess && scratch != 0) {
        // the row loads are in flight in registers the compiler knows nothing about: a build that spills must not run
        k->error = "generated kernel uses " + std::to_string(scratch) + " bytes of scratch per lane (register spills)";
        (void)hipModuleUnload(k->module);
        k->module = nullptr;
        k->func = nullptr;
        return false;
    }
    if (e != hipSuccess) {
        k->error = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        if (k->module) (void)hipModuleUnload(k->module);
        k->module = nullptr;
        k->func = nullptr;
        return false;
    }
    return true;
}

bool comb_code_cached(const CombKernel *k) {
    if (getenv("PARRM_COMB_NO_CACHE")) return false;
    const std::string name = code_name(k->source);
    struct stat sb;
    return stat((library_dir() + "/kernels/" + name).c_str(), &sb) == 0 ||
           (cache_dir_is_safe(user_cache_dir()) && stat((user_cache_dir() + "/" + name).c_str(), &sb) == 0);
}

const char *comb_error(const CombKernel *k) { return k->error.c_str(); }
void comb_set_error(CombKernel *k, const char *msg) { k->error = msg; }
int comb_reach(const CombKernel *k) { return (k->geom.a_hi - k->geom.a_lo + 2) * k->geom.q; }
// LDS reads per real output of stage A (the phase-major kernel's figure is its number of delta taps)
double comb_reads_per_output(const CombKernel *k) {
    double reads = 0.0;
    for (const Unit &u : k->geom.units) reads += k->geom.c + u.width - 1;
    return reads / k->geom.c * (16.0 * k->geom.c) / static_cast<double>(k->geom.q);
}
int comb_stride(const CombKernel *k) { return k->geom.q * k->geom.fsplit; }  // (the recurrence stride)

// The stride this kernel works best at for a tap mask: the one with the fewest LDS reads per output among the
// strides whose geometry it can take (0: none).  Independent of the generic kernels' choice (the phase-major plan
// may prefer its wrap form or another stride).
int64_t comb_search_stride(const std::vector<int8_t> &tap, int64_t hw) {
    int64_t best_q = 0;
    double best = 1e300;
    for (int64_t q = 80; q <= 352; ++q) {  // (above 176: even strides, rows of half a stride -- make_geom)
        Geom g;
        if (!make_geom(tap, hw, q, &g)) continue;
        double reads = 0.0;
        for (const Unit &u : g.units) reads += g.c + u.width - 1;
        const double cost = reads / g.c * (16.0 * g.c) / static_cast<double>(g.q);  // per real output (g.q: the row length)
        if (cost < best) {
            best = cost;
            best_q = q;
        }
    }
    return best_q;
}

// Launch (the kernel's own element types, mode 0).  Fills a->stretch_len / a->n_stretch (the repair pass needs them).
int launch_comb(const CombKernel *k, FilterArgs *args, hipStream_t stream) {
    FilterArgs &a = *args;
    const Geom &g = k->geom;
    // Stretch length: one workgroup per CU (256 resident on a whole MI355X); a stretch costs its rows plus a prologue of
    // a_hi - a_lo + 2 NR rows and one tap-by-tap evaluation (~ 5 rows' worth).  Long recordings: ~1 M samples.
    const int64_t q = g.q;
    const int64_t prologue_rows = g.a_hi - g.a_lo + 2 * kNR + 8;
    // Equal stretches: a recording is cut into round(length / ~1.6 M) stretches of the same number of rows (the
    // last one a little shorter), not into fixed-size stretches plus a remainder -- 10 M samples in 1 M stretches
    // left a tenth round with half-empty workgroups (8.45 -> 8.25 ms).  5 ... 12 equal stretches per 10 M-sample
    // channel all run within 1.5 % of each other; 6 measured best (every stretch pays one prologue and one drain).
    int64_t target = 1677722;
    if (const char *env = getenv("PARRM_COMB_STRETCH")) target = std::max<int64_t>(atoll(env), q * kNR);
    const int64_t n_cut = std::max<int64_t>(1, (a.out_len + target / 2) / target);
    int64_t rows = std::max<int64_t>(kNR, (((a.out_len + n_cut - 1) / n_cut + q - 1) / q + kNR - 1) / kNR * kNR);
    auto blocks_for = [&](int64_t r) { return a.plan_chans * ((a.out_len + r * q - 1) / (r * q)); };
    if (blocks_for(rows) < 2048 && !getenv("PARRM_COMB_STRETCH")) {
        const int64_t resident = parrm::device_cu_count();  // one workgroup per CU (141 KB of LDS)
        double best = 1e300;
        int64_t best_rows = rows;
        const int64_t max_stretches = std::max<int64_t>(1, a.out_len / (static_cast<int64_t>(4) * kNR * q));
        // At least 8 stretches per channel on long recordings: the cut is planned for the WHOLE recording (a channel
        // block of a sharded run must be cut the same way to return the same bits), so one stretch per channel --
        // optimal for 256 channels on one GPU -- would leave a 32-channel shard of an 8-GPU run with 32 workgroups
        // for 256 CUs.  5 ... 12 equal stretches per channel measure within 1.5 % of each other.
        const int64_t min_stretches = std::min<int64_t>(8, std::max<int64_t>(1, a.out_len / (int64_t{1} << 20)));
        for (int64_t per_chan = std::min(min_stretches, max_stretches); per_chan <= std::min<int64_t>(max_stretches, 4096); ++per_chan) {
            const int64_t r = ((a.out_len + per_chan - 1) / per_chan + q - 1) / q;
            const int64_t rr = (r + kNR - 1) / kNR * kNR;
            const int64_t blocks = blocks_for(rr);
            const int64_t rounds = (blocks + resident - 1) / resident;
            const double cost = static_cast<double>(rounds) * static_cast<double>(rr + prologue_rows);
            if (cost < best * (1.0 - 1e-9)) {
                best = cost;
                best_rows = rr;
            }
            if (blocks > 4096) break;
        }
        rows = best_rows;
    }
    a.stretch_len = rows * q;
    a.n_stretch = (a.out_len + a.stretch_len - 1) / a.stretch_len;
    const int64_t blocks = a.n_chans * a.n_stretch;
    PARRM_REQUIRE(blocks <= 0x7fffffffLL, "filter: too many workgroups for one launch");
    CombArgs ca{};
    ca.x = a.x;
    ca.y = a.y;
    ca.n_chans = a.n_chans;
    ca.buf_first = a.buf_first;
    ca.buf_len = a.buf_len;
    ca.out_first = a.out_first;
    ca.out_len = a.out_len;
    ca.n_total = a.n_total;
    ca.ldx = a.ldx;
    ca.ldy = a.ldy;
    ca.tapcum = a.tapcum;
    ca.stretch_len = a.stretch_len;
    ca.n_stretch = a.n_stretch;
    ca.inv_taps = a.inv_taps;
    ca.hw = a.hw;
    size_t sz = sizeof(ca);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ca, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    PARRM_HIP_CHECK(hipModuleLaunchKernel(k->func, static_cast<unsigned>(blocks), 1, 1, static_cast<unsigned>(g.threads), 1, 1, 0, stream, nullptr, config));
    return PARRM_OK;
}

// Can this launch take the comb kernel?  (31-bit byte offsets; the element types are the caller's match.)
bool comb_accepts(const CombKernel *k, const FilterArgs &a) {
    return a.buf_len * (k->in32 ? 4 : 8) < 0x7ffff000LL && a.out_len * (k->out32 ? 4 : 8) < 0x7ffff000LL;
}

}  // namespace parrm_filter

extern "C" {

// Build-time helper (no GPU needed): generate the comb kernel for a filter array (parrm.py:803-833 layout) and
// store its code object as <out_dir>/comb_<hash>.hsaco -- __graft_entry__.build() ships the BASELINE geometry
// that way, so a fresh box does not compile at first use.  `stride` 0: the stride the plan would choose is not
// known without a plan -- pass the phase stride (parrm_filter_plan_info.phase_stride) or let it be searched.
int parrm_filter_comb_precompile(const double *h_filter, int64_t filter_len, int64_t stride, const char *out_dir,
                                 char *source_path, size_t source_path_len) {
    PARRM_REQUIRE(h_filter && out_dir, "comb_precompile: NULL argument");
    PARRM_REQUIRE(filter_len >= 3 && (filter_len & 1), "comb_precompile: filter length must be odd and >= 3");
    const int64_t hw = (filter_len - 1) / 2;
    std::vector<int8_t> tap(filter_len, 0);
    for (int64_t i = 0; i < filter_len; ++i)
        if (i != hw && h_filter[i] != 0.0) tap[i] = 1;
    if (stride == 0) stride = parrm_filter::comb_search_stride(tap, hw);
    // (element types: float64 -> float64 unless PARRM_COMB_PRECOMPILE_TYPES says "f32f64" or "f32f32")
    const char *types = getenv("PARRM_COMB_PRECOMPILE_TYPES");
    const bool in32 = types && strncmp(types, "f32", 3) == 0, out32 = in32 && strcmp(types, "f32f32") == 0;
    parrm_filter::CombKernel *k = stride ? parrm_filter::comb_generate(tap, hw, stride, 0, in32, out32) : nullptr;
    if (!k) {
        parrm::set_error("comb_precompile: the generated kernel does not take this filter");
        return PARRM_ERR_INVALID;
    }
    std::vector<char> code;
    std::string log;
    const bool ok = parrm_filter::compile_source(k->source, &code, &log);
    const std::string name = parrm_filter::code_name(k->source);
    int rc = PARRM_OK;
    if (!ok) {
        parrm::set_error("comb_precompile: %s", log.c_str());
        rc = PARRM_ERR_HIP;
    } else {
        parrm_filter::make_dirs(out_dir);
        if (!parrm_filter::write_file_atomic(std::string(out_dir) + "/" + name, code)) {
            parrm::set_error("comb_precompile: cannot write %s/%s", out_dir, name.c_str());
            rc = PARRM_ERR_INVALID;
        }
        if (source_path && source_path_len) {
            const std::string sp = std::string(out_dir) + "/" + name.substr(0, name.size() - 6) + ".hip";
            std::ofstream f(sp);
            f << k->source;
            snprintf(source_path, source_path_len, "%s", sp.c_str());
        }
    }
    parrm_filter::comb_destroy(k);
    return rc;
}

}  // extern "C"
