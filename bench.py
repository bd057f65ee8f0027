#!/usr/bin/env python3
"""Headline benchmark: Msamples/s through find_period + filter_data (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ONE 256 ch x
10 Msample float64 synthetic recording (22 kHz sampling, 130 Hz artefact) resident in
HBM before the timed region.  One *step* = one full pass of the hot path over it:

    find_period over a ~1e4-point period grid (26 assumed periods, 2 % apart -> 10 044 candidates
    in stage 1, then the reference's stages 2-3 and the final polish)  ->  create_filter()
    (defaults: hw 2372, 196 taps)  ->  filter_data()

Multi-GPU (configs[3]): one process per GPU.  Default ``--mode strong``: the SAME recording,
channel-sharded (rank r builds and holds rows ``channel_shard(256, r, N)``); ``filter_data`` and the
statistics pass run on the rank's rows with no exchange, the candidate grids are cut into per-rank
slices over a replicated stage matrix (``pyparrm_amd.sharding.ShardedPARRM``: two small all-gathers
per stage on the launcher's process group -- the path's one real exchange step), the Nelder-Mead
chains run redundantly.  ``value`` = 256 x 10 M samples per step / time: strong scaling.
``--mode weak``: one independent 256 x 10 M recording and one ``PARRM`` per rank (per-site period
estimation, examples/plot_example_dbs_data.py:52-98), no exchange at all.
Started without a launcher (no WORLD_SIZE) and ``--gpus N > 1``, this script starts the N rank
processes itself -- before it touches the GPU -- and exits with their status.

The JSON line also carries
  roofline      the filter_data kernel against HBM: algorithmic 16 B/sample x
                C x N per launch / the launch's duration from HIP events on its stream;
  cpu_baseline  ``oracle/cpu_baseline.py`` (the reference's own formulation on the host cores: configs[0]
                and configs[1] in full, configs[2] on 8 of its 256 channels), run as a child process
                before the GPU is touched -- rank 0, N=1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS, F_ART = 22000.0, 130.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
FILTER_BYTES_PER_SAMPLE = 16  # f64 in + f64 out


def assumed_periods_1e4():
    base = FS / F_ART
    return tuple(base * (1 + 0.02 * k) for k in range(-13, 13))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chans", type=int, default=256)
    ap.add_argument("--samples", type=int, default=10_000_000)
    ap.add_argument("--mode", choices=["strong", "weak"], default="strong",
                    help="N > 1: one recording channel-sharded (strong) or one recording per rank (weak)")
    ap.add_argument("--workload", choices=["cfg3", "cfg5"], default="cfg3",
                    help="cfg3 (default): the headline, resident f64 find_period + filter_data; cfg5: BASELINE "
                         "configs[4], float32 recording streamed from page-locked host memory, filter_data only")
    ap.add_argument("--exchange", choices=["ipc", "shm", "device", "host"], default="ipc",
                    help="strong mode under a launcher: the two exchanges per stage as device-to-device peer copies through "
                         "IPC memory handles (default: no RCCL in the data path; falls back to `shm`, then `device`), staged "
                         "through page-locked POSIX shared memory, on the process group's device collective (RCCL over xGMI "
                         "with nccl), or through a gloo group")
    ap.add_argument("--spawn", action="store_true",
                    help="no launcher and --gpus N > 1: start N rank processes (default: ONE process, one host thread "
                         "per device -- sharding.MultiDevicePARRM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--filter-only", action="store_true", help="time filter_data alone (config 2 style)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as child processes.  Nothing in
    this process has touched the GPU yet (no HIP call; counting devices does not initialise the runtime),
    and nothing will.  A rank that dies takes the others with it -- they would wait for it forever."""
    if not args.single_device:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    while procs:
        time.sleep(0.2)
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and status == 0:
                status = code
                for other in procs:  # the exact processes started above
                    other.terminate()
    return status


def run_cpu_baseline(save_path=None):
    """The oracle on the host cores, as a child process that never sees the GPU.  ``save_path``: the 8 channels
    of its configs[2] subsample are written there, and the GPU recording takes them as its first 8 channels, so
    that both legs work on the same samples."""
    cmd = [sys.executable, "-m", "oracle.cpu_baseline"] + (["--save-cfg3", save_path] if save_path else [])
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    if res.returncode != 0:
        return {"error": res.stderr[-400:]}
    out = json.loads(res.stdout.strip().splitlines()[-1])
    return {k: out[k] for k in ("value", "unit", "cores", "kind", "sample", "period", "configs", "wall_s")}


def default_taps(period, n_samples):
    """Tap offsets of the default filter for a period (parrm.py:788-833), host arithmetic only."""
    from pyparrm_amd import PARRM

    p = PARRM(np.zeros((1, n_samples)), FS, F_ART, verbose=False)
    p._period = np.float64(period)
    p.create_filter()
    f = p.filter
    return np.flatnonzero((f != 0) & (np.arange(f.size) != f.size // 2))


def launch_ms(fn, reps=5):
    """Mean duration of `fn` (one kernel launch sequence on torch's current stream) by HIP events, one warm-up."""
    import torch

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run_cfg5(args, rank, local_rank, world):
    """BASELINE configs[4]: 1024 ch x 50 Msample float32, filter_data only, streamed in host-pinned
    chunks; channel blocks per GPU (one process each), no exchange.  The link, not the kernel, is the
    bound: reports GB/s over PCIe per GPU next to the metric.  ``--chans/--samples`` size the WHOLE
    recording (defaults here: 1024 x 50 M; one GPU of an 8-GPU run holds 128 channels)."""
    import torch

    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.sharding import channel_shard, timed_steps

    _hip.require_gpu()
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    n_chans = args.chans if args.chans != 256 else 1024
    n_samples = args.samples if args.samples != 10_000_000 else 50_000_000
    lo, hi = channel_shard(n_chans, rank, world)
    rows = hi - lo
    # this rank's block, generated on the device in pieces and parked in page-locked host memory
    x = torch.empty((rows, n_samples), dtype=torch.float32).pin_memory()
    y = torch.empty((rows, n_samples), dtype=torch.float32).pin_memory()
    gen = torch.Generator(device="cuda")
    period = FS / F_ART * (1 + 3e-5)
    for c in range(rows):
        gen.manual_seed(5000 + lo + c)
        n = torch.arange(n_samples, dtype=torch.float64, device="cuda")
        u = torch.remainder((n + 3.0 * (lo + c)) / period, 1.0)
        art = torch.clamp(1.0 - torch.abs(u - 0.25) * 20.0, min=0.0) - 0.5 * torch.clamp(1.0 - torch.abs(u - 0.35) * 12.0, min=0.0)
        row = torch.randn(n_samples, dtype=torch.float32, device="cuda", generator=gen) + (5.0 * art).to(torch.float32)
        x[c].copy_(row)
        del n, u, art, row
    torch.cuda.synchronize()
    p = PARRM(np.zeros((1, n_samples)), FS, F_ART, verbose=False)
    p._period = np.float64(period)
    p.create_filter()
    plan = _hip.FilterPlan(p.filter, device=local_rank)
    x_np, y_np = x.numpy(), y.numpy()

    def step():
        plan.apply_host(x_np, out_dtype=np.float32, out=y_np)  # synchronous: returns when y is complete

    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(step, args.steps, 0, dist=dist, sync=torch.cuda.synchronize)
    if rank == 0:
        info = plan.info
        per_gpu_bytes = 2.0 * rows * n_samples * 4
        out = {
            "metric": "Msamples/s through filter_data (host-streamed, configs[4])",
            "value": n_chans * n_samples * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{n_chans} ch x {n_samples} samples float32 in page-locked host memory, channel blocks of "
                             f"{rows} rows per GPU, filter_data only (f32 out), streamed in time chunks with a "
                             f"{int(info.half_width)}-sample halo over two streams"),
                "parallelism": f"{world} GPU(s), channel blocks, no exchange",
            },
            "roofline": {"bound": "pcie", "achieved": per_gpu_bytes * args.steps / elapsed / 1e9, "peak": 2 * 63.0,
                         "unit": "GB/s", "frac": per_gpu_bytes * args.steps / elapsed / 1e9 / (2 * 63.0), "traffic": None,
                         "note": "per GPU, host->device + device->host together; PCIe Gen5 x16 is 63 GB/s each way"},
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_threads(args):
    """``python bench.py --gpus N`` without a launcher: ONE process, one host thread per device, the recording's
    channel blocks resident on their devices, peer copies for the two small exchanges per stage
    (``sharding.MultiDevicePARRM``).  Same workload and the same JSON line as the launcher path."""
    import torch

    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.sharding import MultiDevicePARRM, channel_shard
    from pyparrm_amd.synth import synth_recording_device

    _hip.require_gpu()
    have = torch.cuda.device_count()
    if have < args.gpus and not args.single_device:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
    devices = [0] * args.gpus if args.single_device else list(range(args.gpus))
    n_chans, n_samples = args.chans, args.samples
    blocks = []
    for r, dev in enumerate(devices):
        lo, hi = channel_shard(n_chans, r, args.gpus)
        with torch.cuda.device(dev):
            blocks.append(synth_recording_device(n_chans, n_samples, FS, F_ART, seed=0, chan_range=(lo, hi), device=f"cuda:{dev}"))
    assumed = assumed_periods_1e4()

    def sync():
        for dev in set(devices):
            torch.cuda.synchronize(dev)

    def step():
        p = MultiDevicePARRM.from_blocks(blocks, FS, F_ART, verbose=False)
        p.find_period(assumed_periods=assumed, random_seed=44)
        p.create_filter()
        y = p.filter_data()
        sync()
        period = p.period
        del y, p
        return period

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        period = step()
    sync()
    elapsed = time.perf_counter() - t0
    out = {
        "metric": "Msamples/s through find_period+filter_data",
        "value": n_chans * n_samples * args.steps / elapsed / 1e6,
        "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": (f"{n_chans} ch x {n_samples} samples float64, 22 kHz / 130 Hz synthetic DBS; find_period over "
                         f"{len(PARRM._get_possible_periods(assumed, 1))} stage-1 candidates (26 assumed periods) + stages 2-3 + polish, "
                         "create_filter defaults, filter_data"),
            "parallelism": (f"one recording, channel blocks on {args.gpus} devices of ONE process (one host thread per device, "
                            "peer copies for the two exchanges per stage, no process group, no RCCL), Nelder-Mead replicated"),
        },
        "period": float(period),
    }
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.spawn or args.workload != "cfg3" or args.mode != "strong":
            raise SystemExit(spawn_ranks(args))
        return run_threads(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.workload == "cfg5":
        return run_cfg5(args, rank, local_rank, world)

    cpu = None
    shared_path = None
    if world == 1 and not args.no_cpu_baseline:
        if args.chans >= 8 and args.samples == 10_000_000:
            shared_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"parrm_bench_cfg3_{os.getpid()}.npy")
        cpu = run_cpu_baseline(shared_path)  # before this process initialises the GPU

    import torch

    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.sharding import ShardedPARRM, TorchExchange, channel_shard, timed_steps
    from pyparrm_amd.synth import synth_recording_device

    _hip.require_gpu()
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # nccl == RCCL

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    n_chans, n_samples = args.chans, args.samples
    strong = world > 1 and args.mode == "strong"
    if strong:
        lo, hi = channel_shard(n_chans, rank, world)
        x = synth_recording_device(n_chans, n_samples, FS, F_ART, seed=0, chan_range=(lo, hi))
        exchange_name, exchange = args.exchange, None
        if exchange_name == "ipc":
            from pyparrm_amd.sharding import IpcExchange

            exchange = IpcExchange.create(dist)  # None on EVERY rank if any rank cannot use IPC handles
            if exchange is None:
                if rank == 0:
                    print("bench.py: IPC-handle exchange unavailable; using shared memory", file=sys.stderr)
                exchange_name = "shm"
        if exchange_name == "shm":
            try:
                from pyparrm_amd.sharding import ShmExchange

                exchange = ShmExchange(dist)
            except Exception as exc:  # e.g. /dev/shm not writable: the device collective is the fallback
                print(f"bench.py: shared-memory exchange unavailable ({exc}); using the device collective", file=sys.stderr)
                exchange_name = "device"
        if exchange is None:
            exchange = TorchExchange(dist, via_host=exchange_name == "host")
    else:
        x = synth_recording_device(n_chans, n_samples, FS, F_ART, seed=1000 * rank)
    shared_rows = 0
    if shared_path and os.path.exists(shared_path):
        # the channels the CPU baseline just worked on become the first channels of the GPU's recording
        head = np.load(shared_path, mmap_mode="r")
        if head.shape[1] == n_samples and head.shape[0] <= x.shape[0]:
            shared_rows = head.shape[0]
            x[:shared_rows] = torch.from_numpy(np.array(head)).to(x.device)
        del head
        os.unlink(shared_path)
    torch.cuda.synchronize()
    assumed = assumed_periods_1e4()

    timings = {"find": [], "filter_total": [], "filter_kernel": [], "grids": []}
    state = {}

    def step(record: bool):
        p = ShardedPARRM(x, FS, F_ART, exchange, verbose=False) if strong else PARRM(x, FS, F_ART, verbose=False)
        t0 = time.perf_counter()
        _hip.FIT_GRID_EVENTS = []
        if args.filter_only and "period" in state:
            p._period = state["period"]
        else:
            p.find_period(assumed_periods=assumed, random_seed=44)
        t1 = time.perf_counter()
        grids, _hip.FIT_GRID_EVENTS = _hip.FIT_GRID_EVENTS, None
        p.create_filter()
        _hip.filter_kernel_timing(True)  # HIP events right around the main kernel, on its stream (not the repair pass)
        y = p.filter_data()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        kernel_ms = _hip.filter_kernel_timing(True)
        state.update(period=p.period, filt=p.filter, info=p._plan.info, generated=p._plan.generated)
        del y, p  # the output goes back to torch's caching allocator for the next step
        if record:
            # candidate grids: bracketed duration and algorithmic flops 2 n K (C + K) per candidate
            timings["grids"].append([(e0.elapsed_time(e1), 2.0 * n * (2 * bw + 1) * (c + 2 * bw + 1) * npd, npd, 2 * bw + 1)
                                     for e0, e1, n, c, npd, bw in grids])
            timings["find"].append(t1 - t0)
            timings["filter_total"].append(t2 - t1)
            timings["filter_kernel"].append(kernel_ms * 1e-3)

    # W untimed warm-up steps, then exactly K steps between two sync+barrier fences; MAX over ranks
    for _ in range(args.warmup):
        step(False)
    elapsed = timed_steps(lambda: step(True), args.steps, 0, dist=dist, sync=torch.cuda.synchronize)

    if rank == 0:
        rows = x.shape[0]
        total_samples = (1 if strong else world) * n_chans * n_samples * args.steps
        kern_s = float(np.mean(timings["filter_kernel"]))
        achieved = FILTER_BYTES_PER_SAMPLE * rows * n_samples / kern_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get("chans") == rows and rec.get("samples") == n_samples:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        info = state["info"]
        if world == 1:
            parallelism = "1 GPU"
        elif strong:
            parallelism = (f"one recording, channel blocks of {rows} rows on {world} GPUs; filter_data and the statistics "
                           "pass without exchange, candidate grids in per-rank slices over a replicated stage matrix "
                           f"(2 exchanges per stage, {'device-to-device peer copies through IPC memory handles, no RCCL in the data path' if exchange_name == 'ipc' else 'staged through page-locked POSIX shared memory, no RCCL in the data path' if exchange_name == 'shm' else 'staged through host memory (gloo)' if exchange_name == 'host' or args.backend != 'nccl' else 'RCCL all_gather over xGMI'}), "
                           "Nelder-Mead replicated")
        else:
            parallelism = f"{world} independent recordings, one per GPU, no exchange"
        out = {
            "metric": "Msamples/s through find_period+filter_data",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.mode,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"{n_chans} ch x {n_samples} samples float64"
                    + (" per GPU" if world > 1 and not strong else "")
                    + ", 22 kHz / 130 Hz synthetic DBS; "
                    + ("filter_data only (period reused)" if args.filter_only else
                       f"find_period over {len(PARRM._get_possible_periods(assumed, 1))} stage-1 candidates "
                       "(26 assumed periods) + stages 2-3 + polish, create_filter defaults, filter_data")
                ),
                "parallelism": parallelism,
                "filter": {"half_width": int(info.half_width), "taps": int(info.n_taps),
                           "stride_q": int(info.phase_stride or info.stride),
                           "delta_taps": int(info.phase_delta or info.n_delta),
                           "row_groups": int(info.phase_groups), "rows_per_thread": int(info.phase_rows)},
            },
            "roofline": {
                "kernel": ("parrm_comb_kernel (generated for this filter by hipRTC, stride %d)" % state["generated"][1]) if state["generated"][0] == 1 else
                          {1: "filter_gather_kernel", 2: "filter_stride_kernel", 3: "filter_phase_kernel", 4: "filter_phase_kernel (segmented plan)"}[int(info.kernel)] + "<double,double>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                                   "scripts/profile_filter.py on this kernel and shape (FETCH_SIZE x 2, the gfx950 correction); "
                                   "a tracked measurement, NOT collected in this run") if traffic is not None else None,
                "launch_ms": kern_s * 1e3,
                "rows_per_launch": rows,
            },
            "breakdown_ms": {
                "find_period": float(np.mean(timings["find"])) * 1e3,
                "create_filter+filter_data": float(np.mean(timings["filter_total"])) * 1e3,
                "filter_kernel": kern_s * 1e3,
            },
            "period": float(state["period"]),
        }
        # ---- second roofline: the candidate grids of find_period (matrix-core bound, 78.6 TFLOP/s FP64)
        if timings["grids"] and timings["grids"][0]:
            per_step_ms = float(np.mean([sum(g[0] for g in gs) for gs in timings["grids"]]))
            flops = float(np.mean([sum(g[1] for g in gs) for gs in timings["grids"]]))
            out["roofline_find_period"] = {
                "kernel": "fit_accum_fused_kernel (bracket = one grid call: design rows + Gram blocks in one kernel, reduction, solves)",
                "bound": "fp64-mfma (deliberate: north_star says 'no MFMA', but the candidate grids are a dense float64 "
                         "contraction W'[Y W] -- SURVEY.md 7 hard part 4 -- and run on the matrix cores; filter_data, the "
                         "bandwidth-bound part, uses none)",
                "flops": flops, "ms_per_step": per_step_ms,
                "achieved": flops / per_step_ms / 1e9, "peak": 78.6, "unit": "TFLOP/s", "frac": flops / per_step_ms / 1e9 / 78.6,
                "grids": [{"candidates": g[2], "K": g[3], "ms": g[0]} for g in timings["grids"][-1]],
                "note": "algorithmic flops 2 n K (C + K) per candidate (SURVEY.md 8d); Nelder-Mead batches are not in the bracket",
            }
        # ---- other configurations and dtypes, untimed region: a few launches each
        if world == 1 and not args.filter_only:
            extra = []
            plan = _hip.FilterPlan(state["filt"])

            def roof(config, kernel, ms, bytes_per_sample, n_c, n_s):
                gbs = bytes_per_sample * n_c * n_s / ms / 1e6
                extra.append({"config": config, "kernel": kernel, "ms": ms, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})

            def kernel_name(pl):
                return "parrm_comb_kernel (generated)" if pl.generated[0] == 1 else "filter_phase_kernel"

            # the box's own copy rate on the headline launch's buffers, in this process: torch's vectorised device copy
            # moves the same 16 B/sample (read 8, write 8) -- the roofline fraction against the 8 TB/s spec says how far
            # the kernel is from the data sheet, this one how far it is from what THIS box copies at
            y_copy = torch.empty_like(x)
            ms_copy = launch_ms(lambda: y_copy.copy_(x), 3)
            del y_copy
            out["roofline"]["copy_ms_same_box"] = ms_copy
            out["roofline"]["frac_of_copy_rate"] = ms_copy / (kern_s * 1e3)
            x2 = x[:64, :1_000_000].contiguous()
            y2 = torch.empty_like(x2)
            ms2 = launch_ms(lambda: plan.apply(x2, out=y2))
            roof("configs[1]: 64 ch x 1 M float64, filter_data only", kernel_name(plan), ms2, 16, 64, 1_000_000)
            del x2, y2
            p32 = _hip.FilterPlan(state["filt"])
            x32 = x.to(torch.float32)
            y32 = torch.empty_like(x32)

            def name32(pl, out_t):
                if pl.generated[0] == 1:
                    return f"parrm_comb_kernel (generated, float32 widened into a float64 ring, {out_t} out)"
                return f"filter_phase_kernel<float,{out_t}>" + (" (packed)" if out_t == "float" else "")

            ms32 = launch_ms(lambda: p32.apply(x32, out=y32, out_dtype=torch.float32), 3)
            roof(f"{rows} ch x {n_samples} float32 -> float32 (configs[4]'s dtype, resident)", name32(p32, "float"), ms32, 8, rows, n_samples)
            del y32
            y64 = torch.empty((rows, n_samples), dtype=torch.float64, device=x.device)
            ms64 = launch_ms(lambda: p32.apply(x32, out=y64), 3)
            roof(f"{rows} ch x {n_samples} float32 -> float64 (the reference's dtype rule)", name32(p32, "double"), ms64, 12, rows, n_samples)
            del x32, y64
            # other sampling geometries on the same 256 ch x 10 M float64 buffers: the reference's own recordings are 1 kHz /
            # 130 Hz (T = 7.69) and 200 Hz / 150 Hz; a 30 kHz recording has T = 231.  Outside the generated kernel's reach
            # (short periods: teeth at every residue of a stride; T > 176: a row does not fit 16 lanes x 11 residues): the
            # phase-major kernel's wrap / guarded forms
            yg = torch.empty_like(x)
            for fs_g, fa_g in ((1000.0, 130.0), (30000.0, 130.0)):
                pg = PARRM(np.zeros((1, n_samples)), fs_g, fa_g, verbose=False)
                pg._period = np.float64(fs_g / fa_g * (1 + 3e-5))
                pg.create_filter()
                plan_g = _hip.FilterPlan(pg.filter)
                ms_g = launch_ms(lambda: plan_g.apply(x, out=yg), 3)
                roof(f"{rows} ch x {n_samples} float64, {fs_g:.0f} Hz sampling of {fa_g:.0f} Hz stimulation (T = {fs_g / fa_g:.2f}, {int(plan_g.info.n_taps)} taps)",
                     kernel_name(plan_g), ms_g, 16, rows, n_samples)
            del yg
            out["extra"] = {"rooflines": extra}
        if cpu is not None:
            if shared_rows and "period" in cpu:
                cpu["shared_channels"] = shared_rows
                cpu["taps_equal"] = bool(np.array_equal(default_taps(cpu["period"], n_samples), default_taps(state["period"], n_samples)))
                cpu["note"] = (f"the {shared_rows} channels of the configs[2] subsample are channels 0-{shared_rows - 1} of the GPU's "
                               "recording; taps_equal compares the default filters of the two periods (CPU: those 8 channels, GPU: all 256)")
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
