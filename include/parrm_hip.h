/*
 * parrm_hip.h -- C ABI of the MI355X (gfx950) PARRM hot path.
 *
 * Drop-in boundary for neuromodulation/PyPARRM v1.2.0dev (pure Python; there is no FFI in
 * the reference, so each entry point below names the reference expression it replaces,
 * file:line relative to /root/reference, `parrm.py` = src/pyparrm/parrm.py).
 *
 * Conventions
 *   - every `d_*` pointer is a DEVICE pointer valid on the current HIP device; `h_*` is host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Nothing here
 *     synchronises the stream unless its comment says so.
 *   - recordings are row-major [n_chans][n_samples] with a row stride `ld` in ELEMENTS
 *     (the reference's C-order ndarray, parrm.py:102-110).
 *   - every function returns PARRM_OK (0) or a PARRM_ERR_* code; the message of the last
 *     failure on the calling thread is available from parrm_hip_last_error().  No
 *     exceptions cross this boundary; the Python facade turns codes into exceptions.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point
 *     fails with PARRM_ERR_NO_DEVICE / PARRM_ERR_HIP.
 */
#ifndef PARRM_HIP_H
#define PARRM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARRM_HIP_ABI_VERSION 3

#define PARRM_OK 0
#define PARRM_ERR_INVALID 1   /* bad argument (NULL, negative size, unsupported dtype...) */
#define PARRM_ERR_HIP 2       /* a HIP runtime call failed                                */
#define PARRM_ERR_NO_DEVICE 3 /* no HIP device visible                                    */
#define PARRM_ERR_WORKSPACE 4 /* caller-provided workspace too small                      */
#define PARRM_ERR_EMPTY_FILTER 5 /* filter has no taps (parrm.py:822-827 raises before this) */
#define PARRM_ERR_INTERNAL 6  /* a self-check of the library failed (e.g. the host's replay of a device-side refinement) */

/* element types of a recording */
#define PARRM_F32 0
#define PARRM_F64 1

/* filter kernel variants (parrm_filter_plan_set_kernel) */
#define PARRM_KERNEL_AUTO 0
#define PARRM_KERNEL_GATHER 1 /* one thread per output, taps gathered from global/L2        */
#define PARRM_KERNEL_STRIDE 2 /* LDS ring + stride-q running-sum recurrence (any tap pattern) */
#define PARRM_KERNEL_PHASE 3  /* phase-major LDS ring, R rows per thread (comb filters, T>=64) */
#define PARRM_KERNEL_SEGMENTED 4 /* reported only: half-width beyond any LDS ring -- the taps are cut into
                                    offset windows, one phase-kernel pass each, summed in float64 */

int parrm_hip_abi_version(void);
const char *parrm_hip_last_error(void);
/* number of visible HIP devices (0 and PARRM_OK when there is none) */
int parrm_hip_device_count(int *count);
/* Release what the library keeps for the life of the process (the page-locked hand-off blocks of
 * parrm_fit_errors_host).  Call it once no other call is in flight and while the HIP runtime is
 * still up -- e.g. from the host language's exit hook, not from a static destructor.  Everything is
 * re-created on demand if the library is used again afterwards.  Plans, workspaces and streams
 * belong to the caller and are not touched. */
int parrm_hip_shutdown(void);

/* ------------------------------------------------------------------------------------
 * filter_data  (parrm.py:835-875; arithmetic at :861-869)
 *
 *   y[c,n] = x[c,n] - mean{ x[c,n-w] : w in taps, 0 <= n-w < n_total }, 0 where no tap is
 *   in range, 0 where the result is not finite (:869).
 *
 * A plan is built from the reference's dense filter array `PARRM._filter`
 * (parrm.py:803-833: 1 at the centre, -1/S on taps, 0 elsewhere; length 2*hw+1).  Building
 * the plan derives the tap runs, picks the recurrence stride q and uploads the tables to
 * the current device (synchronous).  A plan is bound to the device it was created on:
 * parrm_filter_apply* fail with PARRM_ERR_INVALID when another device is current.
 * ------------------------------------------------------------------------------------ */
typedef struct parrm_filter_plan parrm_filter_plan;

typedef struct parrm_filter_plan_info {
    int64_t half_width;  /* hw                                                         */
    int64_t n_taps;      /* S                                                          */
    int64_t n_runs;      /* maximal runs of consecutive taps                           */
    int64_t stride;      /* q of the stride recurrence (0 when the variant is unusable)*/
    int64_t n_delta;     /* signed taps of d_q = tap(.+q) - tap(.)                     */
    int64_t ring_len;    /* LDS ring length in samples                                 */
    int64_t rows_per_fill; /* rows of q samples loaded per fill                        */
    int32_t block_threads;
    int32_t kernel;      /* variant AUTO resolves to for a large recording             */
    int32_t phase_stride;    /* q of the phase-major kernel (0 = unavailable)          */
    int32_t phase_delta;     /* its padded delta taps (both signs)                     */
    int32_t phase_guard;     /* mirrored residues per row side                         */
    int32_t phase_groups;    /* row groups sharing one LDS ring                        */
    int32_t phase_rows;      /* rows per thread per iteration                          */
    int32_t phase_row_slots; /* ring length in rows                                    */
    int32_t phase_residues;  /* adjacent residues per lane: 1, or 3 (the three-residue form) */
    int32_t reserved;        /* passes of a segmented plan (kernel == PARRM_KERNEL_SEGMENTED), else 0 */
} parrm_filter_plan_info;

int parrm_filter_plan_create(const double *h_filter, int64_t filter_len,
                             parrm_filter_plan **plan);
int parrm_filter_plan_destroy(parrm_filter_plan *plan);
int parrm_filter_plan_query(const parrm_filter_plan *plan, parrm_filter_plan_info *info);
int parrm_filter_plan_set_kernel(parrm_filter_plan *plan, int kernel);

/* Whole recording resident on the device: x[n_chans][n_samples] -> y[n_chans][n_samples].
 * x_dtype in {PARRM_F32, PARRM_F64}; y_dtype PARRM_F64 (the reference's dtype rule: float32
 * in -> float64 out) or PARRM_F32 (build option, only with x_dtype PARRM_F32).  x and y
 * must not overlap.  Arithmetic: float64 throughout for float64 output; for float32 output, launches
 * of >= 2^25 samples whose filter the generated kernel takes are float64 arithmetic rounded once (within
 * 1.2e-7 of the largest output), all others sum a row's tap values in float32 (within 6e-6);
 * PARRM_F32_PACKED=1 selects the latter always. */
int parrm_filter_apply(const parrm_filter_plan *plan, const void *d_x, int x_dtype, void *d_y,
                       int y_dtype, int64_t n_chans, int64_t n_samples, int64_t ldx,
                       int64_t ldy, void *stream);

/* Time-window form for chunk streaming: d_x holds samples [buf_first, buf_first+buf_len) of
 * a recording of n_total samples (row stride ldx); outputs [out_first, out_first+out_len)
 * are written to d_y[c*ldy + (n - out_first)].  The window must contain
 * [max(out_first-hw,0), min(out_first+out_len+hw, n_total)). */
int parrm_filter_apply_window(const parrm_filter_plan *plan, const void *d_x, int x_dtype,
                              void *d_y, int y_dtype, int64_t n_chans, int64_t buf_first,
                              int64_t buf_len, int64_t out_first, int64_t out_len,
                              int64_t n_total, int64_t ldx, int64_t ldy, void *stream);

/* Channel-block form (one recording sharded over GPUs by channel blocks, parrm.py:861-866 filters every
 * channel independently): filters `n_chans` channels of a recording that has `total_chans` in all, cutting
 * the time axis exactly as a call on the whole recording would -- so every channel's output is
 * bit-identical to the unsharded call's, whatever the block size.  parrm_filter_apply_window is the
 * total_chans == n_chans case. */
int parrm_filter_apply_block(const parrm_filter_plan *plan, const void *d_x, int x_dtype, void *d_y,
                             int y_dtype, int64_t n_chans, int64_t total_chans, int64_t buf_first,
                             int64_t buf_len, int64_t out_first, int64_t out_len, int64_t n_total,
                             int64_t ldx, int64_t ldy, void *stream);

/* ---- Nelder-Mead refinement of period estimates (a8) --------------------------------------------------------
 * Replaces /root/reference/src/pyparrm/parrm.py:510-517 and :545-550 (scipy.optimize.fmin per start, mapped over
 * the <= 5 best candidates of a stage, and once more for the final polish): SciPy's one-parameter run (rho 1, chi 2,
 * psi 0.5, sigma 0.5, initial simplex {x0, 1.05 x0}, xatol = fatol as given, maxiter = maxfun = 200 when < 0) for
 * several starts in lock-step, the objective evaluated in BATCHES of abscissae: reflection + both contractions of
 * every run in flight (expansion / shrink points in small follow-up batches), and while at most `lookahead_runs`
 * runs are in flight also the points of the likely following step.  Decisions are SciPy's, taken from the same
 * values in the same order; pyparrm_amd/_neldermead.py is the same logic in Python and tests/test_neldermead.py
 * holds the two (and scipy.optimize.fmin) equal. */
typedef struct parrm_nm parrm_nm;
int parrm_nm_create(const double *starts, int n_starts, double xatol, double fatol, int maxiter, int maxfun,
                    int lookahead_runs, parrm_nm **nm);
int parrm_nm_destroy(parrm_nm *nm);
/* Step interface: the next batch of abscissae (ascending; *n = 0: every run has ended), then their values. */
int parrm_nm_next(parrm_nm *nm, double *points, int capacity, int *n);
int parrm_nm_feed(parrm_nm *nm, const double *values, int n);
/* (xopt, fopt, iterations, function calls) of run `run`: fmin(..., full_output=True)[:4]. */
int parrm_nm_result(const parrm_nm *nm, int run, double *xopt, double *fopt, int *iterations, int *funcalls);
/* The whole search on the device objective in one call (replaces scipy.optimize.fmin's chain of dependent
 * evaluations, parrm.py:499-517, :545-550): one parrm_fit_errors_host call per batch, decisions taken on the host
 * between them.  With PARRM_NM_CHAIN=1 (a fresh object, shapes the matrix-core path takes) the refinement runs as a
 * DEVICE-SIDE CHAIN instead: the kernel that closes a batch also takes the state machine's decisions and leaves the
 * next batch in device memory; the host keeps the stream filled, then replays the recorded batches through this
 * object and fails with PARRM_ERR_INTERNAL unless they agree bit for bit.  Same batches, same values either way
 * (tests/test_gpu_parity_r2.py); the chain measured 0.65 ms slower per search on an MI355X and is not the default
 * (profiles/r04_nm_chain_ab.txt).  Workspace:
 * parrm_nm_fit_workspace_bytes.  hist_x / hist_f / batch_sizes (may be NULL): every evaluation, batch by batch. */
size_t parrm_nm_fit_workspace_bytes(int64_t n_idx, int64_t n_chans, int bw);
int parrm_nm_minimise_fit(parrm_nm *nm, const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                          int64_t n_chans, int bw, double lambda, void *d_workspace, size_t workspace_bytes, void *stream,
                          double *hist_x, double *hist_f, int hist_capacity, int *batch_sizes, int batch_capacity,
                          int *n_batches);

/* Counters since the library was loaded: out[0] refinements run as device-side chains, out[1] refinements stepped
 * from the host, out[2] / out[3] the batches of each. */
int parrm_nm_chain_stats(long long out[4]);

/* The state machine the device runs (csrc/parrm_nm_core.h: plain data, no exceptions), compiled for the host: a
 * TEST SURFACE with parrm_nm_next / _feed / _result's contracts (tests/test_neldermead.py holds it equal to the
 * Python generator and to parrm_nm_* above); at most 8 starts. */
int parrm_nmcore_create(const double *starts, int n_starts, double xatol, double fatol, int maxiter, int maxfun,
                        int lookahead_runs, void **handle);
int parrm_nmcore_destroy(void *handle);
int parrm_nmcore_next(void *handle, double *points, int capacity, int *n);
int parrm_nmcore_feed(void *handle, const double *values, int n);
int parrm_nmcore_result(void *handle, int run, double *xopt, double *fopt, int *iterations, int *funcalls);

/* Page-lock / unlock a host buffer for the streamed path below.  parrm_filter_host uses buffers that are
 * already page-locked in place (the fast path) -- the WHOLE range must lie inside one registration -- and
 * stages every other buffer through page-locked buffers of the call's own: it never locks caller memory
 * itself.  Why: in round 3 a range this library had registered, unregistered after the call and registered again
 * for the next call took a GPU memory fault INSIDE the re-registered range, and a later pageable copy through
 * recycled heap pages died the same way (profiles/r03_host_register_fault.txt,
 * profiles/r03_heap_fault_full_suite.txt; ROCm 7.2).  A library-free reproducer of the bare sequence --
 * hipHostRegister -> hipMemcpyAsync -> hipHostUnregister -> hipHostRegister -> hipMemcpyAsync on one 128 MiB
 * malloc, plus kernel access, threads and a recycled address (scripts/exp_host_register_repro.hip) -- ran CLEAN
 * (profiles/r04_host_register_repro.txt): the bare runtime sequence is not sufficient, so the fault is NOT
 * attributed to the runtime; it needed more of this library's round-3 sequence (two neighbouring ranges locked
 * per call around chunked two-stream copies from two threads) and stays unexplained.  Not locking caller
 * memory removes the sequence whichever side owns the defect.  Hence both entry points
 * are NO-OPS unless the environment sets PARRM_HOST_LOCK=1 (which also lets parrm_filter_host lock buffers
 * of >= 64 MiB that are allocations of their own; the first use prints a warning naming the files above): a caller that filters the same recording repeatedly --
 * the reference's parameter explorer re-filters on every widget event (_utils/_plotting.py:568-584) --
 * allocates page-locked memory once (hipHostMalloc, a torch pinned tensor), which both entry points
 * detect and leave alone.  With the opt-in: the buffer must stay allocated until parrm_host_unpin, and
 * nothing else may lock or unlock a range that shares a page with it meanwhile. */
int parrm_host_pin(void *h_ptr, size_t bytes);
int parrm_host_unpin(void *h_ptr);

/* Host-resident recording, streamed through the device in time chunks of `chunk_samples`
 * (0 = pick) on two streams with double buffers; synchronous. */
int parrm_filter_host(const parrm_filter_plan *plan, const void *h_x, int x_dtype, void *h_y,
                      int y_dtype, int64_t n_chans, int64_t n_samples, int64_t ldx,
                      int64_t ldy, int64_t chunk_samples);

/* The filter kernel of float64 recordings is GENERATED per filter geometry (tap offsets as instruction
 * immediates; parrm.py:861-869 is the arithmetic, :803-833 the filter it is specialised for): hipRTC at the
 * first launch of at least 2^25 samples, code objects cached under $PARRM_KERNEL_CACHE (default
 * ~/.cache/pyparrm_amd) and looked up first in <library dir>/kernels/.  This build-time helper needs no GPU:
 * it generates the kernel for a filter array and writes <out_dir>/comb_<hash>.hsaco (and its source, whose
 * path is returned in source_path when that is not NULL).  `stride` 0 = the stride a plan would choose.
 * PARRM_ERR_INVALID when the generated form does not take this filter (such filters run the generic kernels). */
/* Measurement aid: with `enable` != 0 every later parrm_filter_apply* call of the calling thread records HIP events
 * on its stream right around its MAIN kernel (not the non-finite repair pass behind it); `last_ms` (may be NULL)
 * receives the duration of the most recent such kernel, waiting for it to finish (-1 when there is none).
 * enable == 0 releases the events. */
int parrm_filter_kernel_timing(int enable, float *last_ms);

/* Has this plan's generated kernel been used?  state: 0 = no launch large enough yet (or PARRM_COMB=0),
 * 1 = generated, loaded, self-tested against the tap-by-tap kernel and in use for float64 launches,
 * -1 = not available for this filter (message says why; the generic kernels run instead). */
int parrm_filter_plan_generated(const parrm_filter_plan *plan, int *state, int *stride, char *message,
                                size_t message_len);
/* Background mode: a plan whose generated kernel has to be COMPILED (no cached code object; ~1.7 s of hipRTC) hands
 * the build -- compile, load, spill check, self-test -- to a worker thread at the first large launch and serves that
 * launch and the following ones from the generic kernels (same results within the parity bar) until the kernel has
 * passed its self-test; parrm_filter_plan_generated then reports state 2 meanwhile.  For callers that re-filter
 * interactively with ever new filters (the reference's parameter explorer, _utils/_plotting.py:568-584).  Off by
 * default: a channel-sharded run wants every block filtered by the same kernel (bit-identical to the one-device
 * result), which a swap in mid-run would break.  parrm_filter_plan_destroy waits for a build in flight. */
int parrm_filter_plan_set_background(parrm_filter_plan *plan, int on);
int parrm_filter_comb_precompile(const double *h_filter, int64_t filter_len, int64_t stride,
                                 const char *out_dir, char *source_path, size_t source_path_len);

/* ------------------------------------------------------------------------------------
 * find_period, statistics pass  (parrm.py:272-280, `_standardise_data`)
 *
 *   d_scale[c] = mean_i |x[c,i+1] - x[c,i]|            (:274-275)
 * One streaming read of the recording; deterministic two-level reduction.
 * ------------------------------------------------------------------------------------ */
size_t parrm_absdiff_workspace_bytes(int64_t n_chans, int64_t n_samples);
int parrm_absdiff_mean(const void *d_x, int x_dtype, int64_t n_chans, int64_t n_samples,
                       int64_t ldx, double *d_scale, void *d_workspace, size_t workspace_bytes,
                       void *stream);

/* ------------------------------------------------------------------------------------
 * find_period, gather of the standardised columns one stage consumes
 * (parrm.py:274-278 composed with `data_chan[indices]`, :590)
 *
 *   d_y[j*ldy + c] = clip((x[c,idx[j]+1] - x[c,idx[j]]) / d_scale[c], +-outlier_boundary)
 * Sample-major [n_idx][ldy] layout, ldy >= n_chans.  Requires 0 <= idx[j] < n_samples-1.
 * ------------------------------------------------------------------------------------ */
int parrm_gather_standardise(const void *d_x, int x_dtype, int64_t n_chans, int64_t n_samples,
                             int64_t ldx, const int64_t *d_idx, int64_t n_idx,
                             const double *d_scale, double outlier_boundary, double *d_y,
                             int64_t ldy, void *stream);

/* ------------------------------------------------------------------------------------
 * find_period, batched harmonic-regression objective
 * (`_optimise_local` parrm.py:552-597 over `_fit_waves_to_data` :599-632)
 *
 *   for each period T_p:  angles = (idx+1) * (2*pi/T_p)                          (:619)
 *     W = [1, sin(k*angles), cos(k*angles)]_{k=1..bw}                            (:620-623)
 *     beta_c = solve(W'W, W'y_c)   (LU, partial pivoting; exact zero pivot -> +inf) (:626-628)
 *     err_p = mean_c [ mean_j (y_cj - W_j beta_c)^2 + sum_k regu_k beta_ck^2 ]    (:595-597)
 *   regu_k = lambda * k / sum(1..K), K = 2*bw+1                                   (:585-586)
 * d_y is the [n_idx][ldy] matrix written by parrm_gather_standardise.  bw <= 23.
 * ------------------------------------------------------------------------------------ */
size_t parrm_fit_workspace_bytes(int64_t n_idx, int64_t n_chans, int64_t n_periods, int bw);
int parrm_fit_errors(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                     int64_t n_chans, const double *d_periods, int64_t n_periods, int bw,
                     double lambda, double *d_err, void *d_workspace, size_t workspace_bytes,
                     void *stream);

/* Slice form for a candidate grid that is split over several calls (several GPUs: candidates are
 * independent, parrm.py:445-454 maps over them): evaluates `n_periods` candidates of a grid of
 * `grid_periods` candidates with the work split the whole grid would get, so that every candidate's
 * error is bit-identical to what ONE call on the whole grid returns, wherever the grid is cut.
 * parrm_fit_errors is the grid_periods == n_periods case. */
size_t parrm_fit_slice_workspace_bytes(int64_t n_idx, int64_t n_chans, int64_t n_periods,
                                       int64_t grid_periods, int bw);
int parrm_fit_errors_slice(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                           int64_t n_chans, const double *d_periods, int64_t n_periods,
                           int64_t grid_periods, int bw, double lambda, double *d_err,
                           void *d_workspace, size_t workspace_bytes, void *stream);

/* Same computation with the candidate periods and the errors in HOST memory: copies the periods
 * in, runs parrm_fit_errors on `stream`, copies the errors out and synchronises the stream -- one
 * call per optimiser step (the Nelder-Mead phase of parrm.py:510-517,545-550 is a chain of small,
 * latency-bound evaluations).  The workspace must hold parrm_fit_workspace_bytes(...) +
 * 16 * n_periods bytes. */
int parrm_fit_errors_host(const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                          int64_t n_chans, const double *h_periods, int64_t n_periods, int bw,
                          double lambda, double *h_err, void *d_workspace, size_t workspace_bytes,
                          void *stream);

/* Several independent searches advanced together (per-site period estimation: one PARRM per
 * recording site, examples/plot_example_dbs_data.py:52-98; each optimiser step of every site is a
 * handful of candidates on that site's own stage matrix).  Every problem is one
 * parrm_fit_errors_host call's worth of work -- same kernels, same shapes, so each problem's errors
 * are bit-identical to that call's -- but the problems are dealt over side streams of the library,
 * forked from and joined to `stream`, so their small kernels overlap on the device, and the host
 * waits once.  At most 64 problems and 4096 candidates per call; each problem brings its own
 * workspace of parrm_fit_workspace_bytes(...) + 16 * n_periods bytes.  Synchronous. */
typedef struct parrm_fit_problem {
    const double *d_y;       /* [n_idx][ldy] stage matrix of this search                 */
    int64_t ldy;
    const int64_t *d_idx;    /* [n_idx] sample indices                                   */
    int64_t n_idx, n_chans;
    const double *h_periods; /* [n_periods] candidates (host)                            */
    int64_t n_periods;
    int32_t bw;
    int32_t reserved;
    double lambda;
    double *h_err;           /* [n_periods] errors (host, out)                           */
    void *d_workspace;
    size_t workspace_bytes;
} parrm_fit_problem;
int parrm_fit_errors_multi(const parrm_fit_problem *problems, int n_problems, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PARRM_HIP_H */
