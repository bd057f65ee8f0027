// ABI plumbing: error strings, version, device count, and the host-buffer streaming form of
// filter_data (BASELINE config 5: recordings larger than HBM, or simply host-resident).
#include <atomic>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <csignal>
#include <cstring>
#include <execinfo.h>
#include <unistd.h>

#include "parrm_common.h"

namespace parrm {

static thread_local char g_last_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    if (dev >= 0 && dev < 64) {
        const int c = cached[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    if (dev >= 0 && dev < 64) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

int hip_fail(hipError_t err, const char *what) {
    set_error("%s: %s (%s)", what, hipGetErrorString(err), hipGetErrorName(err));
    return err == hipErrorNoDevice ? PARRM_ERR_NO_DEVICE : PARRM_ERR_HIP;
}

}  // namespace parrm

namespace {

constexpr int kMaxLanes = 4;

struct StreamRes {
    hipStream_t stream[kMaxLanes] = {};
    hipEvent_t done[kMaxLanes] = {};
    void *d_in[kMaxLanes] = {};
    void *d_out[kMaxLanes] = {};
    void *h_in[kMaxLanes] = {};   // library-owned page-locked staging (see parrm_filter_host)
    void *h_out[kMaxLanes] = {};
    ~StreamRes() {
        for (int i = 0; i < kMaxLanes; ++i) {
            if (d_in[i]) (void)hipFree(d_in[i]);
            if (d_out[i]) (void)hipFree(d_out[i]);
            if (h_in[i]) (void)hipHostFree(h_in[i]);
            if (h_out[i]) (void)hipHostFree(h_out[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (stream[i]) (void)hipStreamDestroy(stream[i]);
        }
    }
};

}  // namespace

extern "C" {

int parrm_hip_abi_version(void) { return PARRM_HIP_ABI_VERSION; }

const char *parrm_hip_last_error(void) { return parrm::g_last_error; }

int parrm_hip_device_count(int *count) {
    PARRM_REQUIRE(count, "device_count: NULL argument");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) {
        *count = 0;
        return PARRM_OK;
    }
    if (e != hipSuccess) {
        *count = 0;
        return parrm::hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return PARRM_OK;
}

static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t attr{};
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory reports an error here
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Is the WHOLE range [p, p + bytes) page-locked, inside one registration?  (is_pinned_host looks at the first
// byte only: a registration that is shorter than the buffer -- the caller pinned a smaller array that has since
// been freed and reallocated larger, or pinned a neighbour that shares the first page -- would let the copy
// engines run past the locked range.)
static bool is_pinned_range(const void *p, size_t bytes) {
    if (bytes == 0 || !is_pinned_host(p)) return false;
    void *start = nullptr;
    size_t size = 0;
    const hipDeviceptr_t dp = const_cast<void *>(p);
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, dp) == hipSuccess &&
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, dp) == hipSuccess && start && size) {
        return static_cast<const char *>(p) + bytes <= static_cast<const char *>(start) + size;
    }
    (void)hipGetLastError();  // attribute not available: the last byte must at least be page-locked too
    return is_pinned_host(static_cast<const char *>(p) + bytes - 1);
}

// Optional native backtrace when the process aborts (PARRM_ABORT_TRACE=1; tests/conftest.py sets it): an abort() raised
// inside a library this one calls -- the HIP runtime, hipRTC's compiler -- otherwise leaves nothing but Python's own
// stack in the log (round 3 met one such abort in a full test run, once, with no message).  backtrace() and
// backtrace_symbols_fd() write straight to the descriptor; the handler that was installed before is restored and the
// signal re-raised.
namespace {
void (*g_prev_abort_handler)(int) = SIG_DFL;
void abort_trace_handler(int sig) {
    static const char head[] = "\nparrm: SIGABRT -- native backtrace of the aborting thread:\n";
    (void)!write(2, head, sizeof head - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, g_prev_abort_handler == SIG_IGN ? SIG_DFL : g_prev_abort_handler);  // (Python's faulthandler, if it was there)
    raise(sig);
}
struct AbortTrace {
    AbortTrace() {
        if (getenv("PARRM_ABORT_TRACE")) {
            void *warm[4];
            (void)backtrace(warm, 4);  // (loads libgcc's unwinder now, not inside the handler)
            g_prev_abort_handler = signal(SIGABRT, abort_trace_handler);
            if (g_prev_abort_handler == SIG_ERR) g_prev_abort_handler = SIG_DFL;
        }
    }
} g_abort_trace;
}  // namespace

// Optional trace of every lock / unlock this library performs (PARRM_HOST_TRACE=1, stderr): round 2 met a GPU page
// fault at a heap address and could only argue about its cause afterwards -- the next one can be matched to a range.
static void trace_lock(const char *what, const void *p, size_t bytes, int rc) {
    static const bool on = getenv("PARRM_HOST_TRACE") != nullptr;
    if (on) fprintf(stderr, "parrm host: %s [%p, %p) %zu bytes -> %d\n", what, p, static_cast<const char *>(p) + bytes, bytes, rc);
}

// The opt-in that lets this library call hipHostRegister on caller memory.  Read at every call (a test or a caller may
// set it late); the first call that finds it set says on stderr what it risks.
static bool host_lock_opted_in() {
    if (!getenv("PARRM_HOST_LOCK")) return false;
    static std::atomic<bool> warned{false};
    if (!warned.exchange(true))
        fprintf(stderr,
                "pyparrm_amd: PARRM_HOST_LOCK=1 -- the library will hipHostRegister caller memory.  On ROCm 7.2 a host range that "
                "was registered, unregistered and registered again took GPU page faults inside the re-registered range "
                "(profiles/r03_host_register_fault.txt, profiles/r04_host_register_repro.txt); prefer page-locked "
                "ALLOCATIONS (hipHostMalloc, a torch pinned tensor), which the library uses in place.\n");
    return true;
}

// Since the end of round 3 a no-op unless PARRM_HOST_LOCK=1: on this ROCm build a host range that has been through
// hipHostRegister + hipHostUnregister is not safe to lock again -- by anyone.  The runtime itself locks pageable
// memory in place for large hipMemcpy transfers, so after a test had pinned and unpinned two 1-2 MB heap arrays, the
// NEXT test's `torch.from_numpy(x).cuda()` (its array in the same, recycled heap pages) died with "Memory access
// fault by GPU ... on address <heap address>" (profiles/r03_heap_fault_full_suite.txt; three full test runs of the
// round ended that way, at whatever copy came next; the same signature as profiles/r03_host_register_fault.txt).
// Buffers that are not page-locked are staged by parrm_filter_host; callers who want the in-place rate allocate
// page-locked memory once (hipHostMalloc, a torch pinned tensor), which is never unregistered.
int parrm_host_pin(void *h_ptr, size_t bytes) {
    PARRM_REQUIRE(h_ptr && bytes > 0, "host_pin: NULL or empty buffer");
    if (is_pinned_range(h_ptr, bytes)) return PARRM_OK;
    if (!host_lock_opted_in()) return PARRM_OK;
    const hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterDefault);
    trace_lock("pin", h_ptr, bytes, static_cast<int>(e));
    PARRM_HIP_CHECK(e);
    return PARRM_OK;
}

int parrm_host_unpin(void *h_ptr) {
    PARRM_REQUIRE(h_ptr, "host_unpin: NULL buffer");
    if (!host_lock_opted_in() || !is_pinned_host(h_ptr)) return PARRM_OK;  // (never unregisters what it did not register)
    const hipError_t e = hipHostUnregister(h_ptr);
    trace_lock("unpin", h_ptr, 0, static_cast<int>(e));
    PARRM_HIP_CHECK(e);
    return PARRM_OK;
}

// Time-chunked streaming: chunk k covers outputs [k*chunk, (k+1)*chunk) and is uploaded with a
// half-width halo on each side (stencil locality, SURVEY.md 5 "long-context").  Two streams, two
// device buffer pairs: upload/compute/download of chunk k+1 overlaps chunk k.
//
// How the caller's host buffers reach the copy engines, per buffer:
//   * already page-locked (hipHostMalloc, hipHostRegister, parrm_host_pin, a torch pinned tensor): used
//     in place -- the fast path (80+ GB/s both ways);
//   * not locked: the chunks go through page-locked staging buffers the call owns (below).  (With
//     PARRM_HOST_LOCK=1 a buffer of >= 64 MiB that is an allocation of its own is locked in place for the
//     duration of the call instead -- rounds 1-2's default, retired: see usable_in_place.)
//   * staged: the chunks go through page-locked staging
//     buffers the call owns, filled and drained by the calling thread.  A registration covers whole
//     pages and HIP does not count references (a second hipHostRegister of a range succeeds and ONE
//     hipHostUnregister drops the mapping -- scripts/exp_host_register.py), so locking a small buffer
//     that shares its first or last page with another live allocation (glibc serves arrays of up to
//     32 MiB from the heap once its dynamic mmap threshold has grown) is not the library's to do:
//     the round-2 filter fuzz met a GPU page fault at a heap address that way, ~140 cases into a
//     run.  Buffers of >= 64 MiB are always their own mapping.
int parrm_filter_host(const parrm_filter_plan *plan, const void *h_x, int x_dtype, void *h_y, int y_dtype,
                      int64_t n_chans, int64_t n_samples, int64_t ldx, int64_t ldy, int64_t chunk_samples) {
    PARRM_REQUIRE(plan && h_x && h_y, "filter_host: NULL argument");
    PARRM_REQUIRE(x_dtype == PARRM_F32 || x_dtype == PARRM_F64, "filter_host: bad x_dtype %d", x_dtype);
    PARRM_REQUIRE(y_dtype == PARRM_F64 || (y_dtype == PARRM_F32 && x_dtype == PARRM_F32),
                  "filter_host: y_dtype must be f64, or f32 for f32 input");
    PARRM_REQUIRE(n_chans >= 0 && n_samples >= 0 && ldx >= n_samples && ldy >= n_samples, "filter_host: bad shape");
    if (n_chans == 0 || n_samples == 0) return PARRM_OK;
    parrm_filter_plan_info info;
    int rc = parrm_filter_plan_query(plan, &info);
    if (rc != PARRM_OK) return rc;
    const int64_t hw = info.half_width;
    const size_t xs = x_dtype == PARRM_F64 ? 8 : 4, ys = y_dtype == PARRM_F64 ? 8 : 4;
    StreamRes r;
    // chunks in flight: the upload of chunk k+1 overlaps the kernel and the download of chunk k.  Measured
    // (128 ch x 20 M f32 -> f32, scripts/cfg5_sweep.sh): 2 lanes x 256 MiB chunks 82 GB/s over PCIe (in + out);
    // 3 or 4 lanes no better (the link is the limit), 1 GiB chunks 40-60 GB/s (pipeline fill and drain).
    // PARRM_HOST_LANES / PARRM_HOST_CHUNK_MB: tuning knobs.
    int lanes = 2;
    if (const char *env = getenv("PARRM_HOST_LANES")) lanes = std::max(1, std::min(kMaxLanes, atoi(env)));
    const size_t x_bytes = static_cast<size_t>((n_chans - 1) * ldx + n_samples) * xs;
    const size_t y_bytes = static_cast<size_t>((n_chans - 1) * ldy + n_samples) * ys;
    constexpr size_t kLockInPlaceMin = size_t{64} << 20;
    bool locked_x = false, locked_y = false;  // locked by this call: released below
    auto usable_in_place = [&](const void *p, size_t bytes, bool *locked_here) {
        if (is_pinned_range(p, bytes)) return true;  // first AND last byte inside one registration
        if (is_pinned_host(p)) return false;         // partly locked by someone else: neither trusted nor re-locked
        // Locked in place only when the buffer is an allocation of its own: >= 64 MiB (glibc never serves that from
        // the heap) AND it starts within a malloc header of a page boundary.  A row-block VIEW of a larger array
        // (sharding.shard_recording; np.ascontiguousarray does not copy it) starts anywhere inside its parent's
        // mapping: its first and last page belong to the neighbouring blocks as well, two threads filtering two
        // blocks would lock and unlock each other's edge pages (a registration covers whole pages, HIP keeps no
        // reference count) -- such a buffer is staged, or its caller pins the parent array once (parrm_host_pin).
        //
        // Since round 3 the call does not lock caller memory at all unless PARRM_HOST_LOCK=1: a range that had been
        // registered, unregistered and registered AGAIN (the second call on the same array) took a GPU page fault
        // inside the range the second hipHostRegister had reported as locked (profiles/r03_host_register_fault.txt:
        // the trace of every lock next to the faulting address) -- the same signature as round 2's unexplained
        // fault.  Unlocked buffers are staged through page-locked buffers the call owns; callers who need the
        // in-place rate allocate page-locked memory (hipHostMalloc, a torch pinned tensor) once.
        const bool own_mapping = (reinterpret_cast<uintptr_t>(p) & 4095u) <= 128u;
        if (bytes >= kLockInPlaceMin && own_mapping && host_lock_opted_in()) {
            const hipError_t er = hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault);
            trace_lock("lock for the call", p, bytes, static_cast<int>(er));
            if (er == hipSuccess) {
                *locked_here = true;
                return true;
            }
            (void)hipGetLastError();  // refused (lock limit): staged instead
        }
        return false;
    };
    const bool direct_x = usable_in_place(h_x, x_bytes, &locked_x);
    const bool direct_y = usable_in_place(h_y, y_bytes, &locked_y);
    auto unpin = [&]() {
        if (locked_x) trace_lock("unlock", h_x, x_bytes, static_cast<int>(hipHostUnregister(const_cast<void *>(h_x))));
        if (locked_y) trace_lock("unlock", h_y, y_bytes, static_cast<int>(hipHostUnregister(h_y)));
    };
    if (chunk_samples <= 0) {
        // ~256 MiB of input per chunk when both buffers are used in place (PARRM_HOST_CHUNK_MB), 4 MiB when a side is
        // staged (the staging buffers are page-locked allocations of two chunks each, and the calling thread's copies
        // only overlap the device's work chunk by chunk: 8 ch x 1 M float64 8.5 ms at 4 MiB, 12.4 at 16, 17.5 at 256;
        // scripts/bench_host_staged.py); never less than a few half-widths (halo overhead <= ~12%)
        int64_t mb = direct_x && direct_y ? 256 : 4;
        if (const char *env = getenv("PARRM_HOST_CHUNK_MB")) mb = std::max<int64_t>(1, atoll(env));
        chunk_samples = std::max<int64_t>((mb << 20) / static_cast<int64_t>(xs) / n_chans, 16 * hw + 1024);
    }
    chunk_samples = std::min(chunk_samples, n_samples);
    const int64_t n_chunks = (n_samples + chunk_samples - 1) / chunk_samples;
    const int64_t buf_cap = chunk_samples + 2 * hw;
    const int nbuf = static_cast<int>(std::min<int64_t>(n_chunks, lanes));
    hipError_t e = hipSuccess;
    for (int i = 0; i < nbuf && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&r.stream[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&r.d_in[i], static_cast<size_t>(n_chans * buf_cap) * xs);
        if (e == hipSuccess) e = hipMalloc(&r.d_out[i], static_cast<size_t>(n_chans * chunk_samples) * ys);
        if (e == hipSuccess && !direct_x) e = hipHostMalloc(&r.h_in[i], static_cast<size_t>(n_chans * buf_cap) * xs, hipHostMallocDefault);
        if (e == hipSuccess && !direct_y) e = hipHostMalloc(&r.h_out[i], static_cast<size_t>(n_chans * chunk_samples) * ys, hipHostMallocDefault);
    }
    if (e != hipSuccess) {
        unpin();
        return parrm::hip_fail(e, "filter_host: buffers");
    }

    // staged output: the chunk a lane produced last waits in its h_out until the lane is reused (or the end)
    int64_t pend_o0[kMaxLanes] = {}, pend_len[kMaxLanes] = {};
    auto drain = [&](int b) -> hipError_t {
        if (direct_y || pend_len[b] == 0) return hipSuccess;
        const hipError_t es = hipEventSynchronize(r.done[b]);
        if (es != hipSuccess) return es;
        const size_t row = static_cast<size_t>(pend_len[b]) * ys;
        for (int64_t c = 0; c < n_chans; ++c)
            memcpy(static_cast<char *>(h_y) + (static_cast<size_t>(c) * ldy + pend_o0[b]) * ys,
                   static_cast<const char *>(r.h_out[b]) + static_cast<size_t>(c) * row, row);
        pend_len[b] = 0;
        return hipSuccess;
    };

    for (int64_t k = 0; k < n_chunks && e == hipSuccess && rc == PARRM_OK; ++k) {
        const int b = static_cast<int>(k % nbuf);
        const int64_t o0 = k * chunk_samples;
        const int64_t olen = std::min(chunk_samples, n_samples - o0);
        const int64_t b0 = std::max<int64_t>(o0 - hw, 0);
        const int64_t b1 = std::min<int64_t>(o0 + olen + hw, n_samples);
        const int64_t blen = b1 - b0;
        // stream order on r.stream[b] already serialises reuse of the device buffer pair b
        if (direct_x) {
            e = hipMemcpy2DAsync(r.d_in[b], static_cast<size_t>(blen) * xs,
                                 static_cast<const char *>(h_x) + static_cast<size_t>(b0) * xs,
                                 static_cast<size_t>(ldx) * xs, static_cast<size_t>(blen) * xs,
                                 static_cast<size_t>(n_chans), hipMemcpyHostToDevice, r.stream[b]);
        } else {
            // the lane's previous upload has left h_in once its download event has fired
            if (k >= nbuf) e = hipEventSynchronize(r.done[b]);
            if (e != hipSuccess) break;
            const size_t row = static_cast<size_t>(blen) * xs;
            for (int64_t c = 0; c < n_chans; ++c)
                memcpy(static_cast<char *>(r.h_in[b]) + static_cast<size_t>(c) * row,
                       static_cast<const char *>(h_x) + (static_cast<size_t>(c) * ldx + b0) * xs, row);
            e = hipMemcpyAsync(r.d_in[b], r.h_in[b], static_cast<size_t>(n_chans) * row, hipMemcpyHostToDevice, r.stream[b]);
        }
        if (e != hipSuccess) break;
        e = drain(b);  // (after the upload was queued: the copy-out overlaps it)
        if (e != hipSuccess) break;
        rc = parrm_filter_apply_window(plan, r.d_in[b], x_dtype, r.d_out[b], y_dtype, n_chans, b0, blen, o0, olen,
                                       n_samples, blen, olen, r.stream[b]);
        if (rc != PARRM_OK) break;
        if (direct_y) {
            e = hipMemcpy2DAsync(static_cast<char *>(h_y) + static_cast<size_t>(o0) * ys, static_cast<size_t>(ldy) * ys,
                                 r.d_out[b], static_cast<size_t>(olen) * ys, static_cast<size_t>(olen) * ys,
                                 static_cast<size_t>(n_chans), hipMemcpyDeviceToHost, r.stream[b]);
        } else {
            e = hipMemcpyAsync(r.h_out[b], r.d_out[b], static_cast<size_t>(n_chans * olen) * ys, hipMemcpyDeviceToHost,
                               r.stream[b]);
            pend_o0[b] = o0;
            pend_len[b] = olen;
        }
        if (e == hipSuccess) e = hipEventRecord(r.done[b], r.stream[b]);
    }
    for (int i = 0; i < nbuf; ++i) {
        const hipError_t e2 = hipStreamSynchronize(r.stream[i]);
        if (e == hipSuccess) e = e2;
    }
    for (int i = 0; i < nbuf && e == hipSuccess && rc == PARRM_OK; ++i) e = drain(i);
    unpin();
    if (rc != PARRM_OK) return rc;
    if (e != hipSuccess) return parrm::hip_fail(e, "filter_host: streaming");
    return PARRM_OK;
}

}  // extern "C"
