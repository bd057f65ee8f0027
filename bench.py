#!/usr/bin/env python3
"""Headline benchmark: Msamples/s through find_period + filter_data (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): per GPU one
256 ch x 10 Msample float64 synthetic recording (22 kHz sampling, 130 Hz artefact)
resident in HBM before the timed region.  One *step* = one full pass of the hot path over it:

    find_period over a ~1e4-point period grid (26 assumed periods, 2 % apart -> 10 044 candidates
    in stage 1, then the reference's stages 2-3 and the final polish)  ->  create_filter()
    (defaults: hw 2372, 196 taps)  ->  filter_data()

Multi-GPU: recordings are independent units (per-site period estimation, one PARRM per rank),
one process per GPU, no data-path collective: weak scaling.  Only the timing uses
torch.distributed (barrier + MAX over ranks).

The JSON line also carries
  roofline      the filter_data kernel against HBM: algorithmic 16 B/sample x
                C x N per launch / the launch's duration from HIP events on its stream;
  cpu_baseline  the oracle (the reference's own formulation: NumPy/LAPACK regression,
                scipy fftconvolve, scipy fmin) timed on this host on ONE channel of the same
                recording (all 10 M samples, same 1e4 grid) -- rank 0, N=1 only.
"""

from __future__ import annotations

import argparse
import json
import os
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--filter-only", action="store_true", help="time filter_data alone (config 2 style)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the timing barrier (nccl = RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def cpu_baseline(channel: np.ndarray):
    """Oracle on one channel of the same recording; returns the cpu_baseline object."""
    from oracle import parrm_oracle as orc

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = 1
    x = np.ascontiguousarray(channel[None, :])
    t0 = time.perf_counter()
    period = orc.find_period(x, FS, F_ART, assumed_periods=assumed_periods_1e4(), random_seed=44)
    t_find = time.perf_counter() - t0
    hw = orc.default_filter_half_width(x.shape[1], period, 0, period / 50)
    filt = orc.generate_filter(period, hw, 0, "both", None)
    t0 = time.perf_counter()
    orc.filter_data_fft(x, filt)
    t_filt = time.perf_counter() - t0
    total = t_find + t_filt
    return {
        "value": x.size / total / 1e6,
        "unit": "Msamples/s",
        "cores": int(threads),
        "kind": "port",
        "sample": (
            f"1 of the recording's channels x {x.shape[1]} samples, same 1e4-candidate find_period "
            f"(oracle: NumPy/LAPACK fits + scipy fmin, {t_find:.1f} s) + filter_data as two scipy "
            f"fftconvolve calls ({t_filt:.1f} s); host has {os.cpu_count()} logical CPUs, BLAS "
            f"threads {threads}; find_period cost is per channel, so Msamples/s scales with N"
        ),
        "period": float(period),
    }, float(period)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.sharding import timed_steps
    from pyparrm_amd.synth import synth_recording_device

    _hip.require_gpu()
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # nccl == RCCL; used for the timing barrier / MAX only

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    n_chans, n_samples = args.chans, args.samples
    x = synth_recording_device(n_chans, n_samples, FS, F_ART, seed=1000 * rank)
    torch.cuda.synchronize()
    assumed = assumed_periods_1e4()

    timings = {"find": [], "filter_total": [], "filter_kernel": []}
    state = {}

    def step(record: bool):
        p = PARRM(x, FS, F_ART, verbose=False)
        t0 = time.perf_counter()
        if args.filter_only and "period" in state:
            p._period = state["period"]
        else:
            p.find_period(assumed_periods=assumed, random_seed=44)
        t1 = time.perf_counter()
        p.create_filter()
        # the filter launch is bracketed by HIP events on the stream it is launched on
        # (pyparrm_amd._hip.FilterPlan.apply records them right around the C-ABI call)
        _hip.FILTER_LAUNCH_EVENTS = []
        y = p.filter_data()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        (ev0, ev1), = _hip.FILTER_LAUNCH_EVENTS
        _hip.FILTER_LAUNCH_EVENTS = None
        state.update(period=p.period, filt=p.filter, info=p._plan.info)
        del y, p  # the 20 GB output goes back to torch's caching allocator for the next step
        if record:
            timings["find"].append(t1 - t0)
            timings["filter_total"].append(t2 - t1)
            timings["filter_kernel"].append(ev0.elapsed_time(ev1) * 1e-3)

    # W untimed warm-up steps, then exactly K steps between two sync+barrier fences; MAX over ranks
    elapsed = timed_steps(lambda: step(True), args.steps, 0, dist=dist, sync=torch.cuda.synchronize) \
        if args.warmup == 0 else None
    if elapsed is None:
        for _ in range(args.warmup):
            step(False)
        elapsed = timed_steps(lambda: step(True), args.steps, 0, dist=dist, sync=torch.cuda.synchronize)

    if rank == 0:
        total_samples = world * n_chans * n_samples * args.steps
        kern_s = float(np.mean(timings["filter_kernel"]))
        achieved = FILTER_BYTES_PER_SAMPLE * n_chans * n_samples / kern_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get("chans") == n_chans and rec.get("samples") == n_samples:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        info = state["info"]
        out = {
            "metric": "Msamples/s through find_period+filter_data",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"{n_chans} ch x {n_samples} samples float64 per GPU, 22 kHz / 130 Hz synthetic DBS; "
                    + ("filter_data only (period reused)" if args.filter_only else
                       f"find_period over {len(PARRM._get_possible_periods(assumed, 1))} stage-1 candidates "
                       "(26 assumed periods) + stages 2-3 + polish, create_filter defaults, filter_data")
                ),
                "parallelism": f"{world} independent recordings, one per GPU, no collectives",
                "filter": {"half_width": int(info.half_width), "taps": int(info.n_taps),
                           "stride_q": int(info.phase_stride or info.stride),
                           "delta_taps": int(info.phase_delta or info.n_delta),
                           "row_groups": int(info.phase_groups), "rows_per_thread": int(info.phase_rows)},
            },
            "roofline": {
                "kernel": {1: "filter_gather_kernel", 2: "filter_stride_kernel", 3: "filter_phase_kernel"}[int(info.kernel)] + "<double,double>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launch_ms": kern_s * 1e3,
            },
            "breakdown_ms": {
                "find_period": float(np.mean(timings["find"])) * 1e3,
                "create_filter+filter_data": float(np.mean(timings["filter_total"])) * 1e3,
                "filter_kernel": kern_s * 1e3,
            },
            "period": float(state["period"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            chan0 = x[0].cpu().numpy()
            base, cpu_period = cpu_baseline(chan0)
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
