#!/usr/bin/env python3
"""Period parity sweep against the reference-written fixtures (tests/golden/r2_periods.json):
runs every case once per trig mode (default: one sincos + angle-addition recurrence; exact:
PARRM_FIT_EXACT_TRIG=1 = sincos(fl(k*a)) per harmonic, the reference's operation order) in its own
process -- the knob is read once per process -- and prints one table.

    python scripts/fuzz_period_r2.py > gpurun_out/r02_fuzz_period.txt
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import numpy as np

    from pyparrm_amd import PARRM, _hip
    from pyparrm_amd.synth import synth_recording_exact

    _hip.require_gpu()
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_periods.json")))
    rows = []
    for group in ("fuzz", "short", "float32", "grid26"):
        for c in cases[group]:
            x = synth_recording_exact(c["n_chans"], c["n_samples"], c["period"], c["seed"],
                                      gain_range=tuple(c["gain"]), dtype=np.dtype(c["dtype"]))
            p = PARRM(x, c["fs"], c["fa"], verbose=False)
            kw = {"assumed_periods": tuple(c["assumed"])} if "assumed" in c else {}
            p.find_period(random_seed=c["random_seed"], **kw)  # warm (plans, workspaces)
            t0 = time.perf_counter()
            p.find_period(random_seed=c["random_seed"], **kw)
            dt = time.perf_counter() - t0
            rows.append({"tag": c["tag"], "ref": c["ref_period"], "got": float(p.period), "ms": dt * 1e3,
                         "ref_s": c["ref_seconds"]})
    print(json.dumps(rows))


def main():
    if os.environ.get("PARRM_FUZZ_WORKER"):
        return worker()
    out = {}
    for mode, env in (("default", {}), ("exact", {"PARRM_FIT_EXACT_TRIG": "1"})):
        e = dict(os.environ, PARRM_FUZZ_WORKER="1", **env)
        res = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, check=True)
        out[mode] = json.loads(res.stdout.strip().splitlines()[-1])
    print("period parity vs the unmodified reference (tests/golden/r2_periods.json); rel = |T - T_ref| / T_ref")
    print(f"{'case':<14}{'T_ref':>22}{'rel default':>14}{'rel exact':>12}{'ms default':>12}{'ms exact':>10}{'reference s':>13}")
    worst = {"default": 0.0, "exact": 0.0}
    ident = {"default": 0, "exact": 0}
    tot = {"default": 0.0, "exact": 0.0}
    for a, b in zip(out["default"], out["exact"]):
        ra, rb = abs(a["got"] - a["ref"]) / a["ref"], abs(b["got"] - b["ref"]) / b["ref"]
        for k, r, row in (("default", ra, a), ("exact", rb, b)):
            worst[k] = max(worst[k], r)
            ident[k] += r == 0.0
            tot[k] += row["ms"]
        print(f"{a['tag']:<14}{a['ref']:>22.15g}{ra:>14.2e}{rb:>12.2e}{a['ms']:>12.1f}{b['ms']:>10.1f}{a['ref_s']:>13.1f}")
    n = len(out["default"])
    print(f"\n{n} cases; bit-identical periods: default {ident['default']}, exact {ident['exact']}; "
          f"worst rel: default {worst['default']:.2e}, exact {worst['exact']:.2e}; "
          f"total find_period time: default {tot['default']:.0f} ms, exact {tot['exact']:.0f} ms "
          f"(reference: {sum(r['ref_s'] for r in out['default']):.0f} s)")


if __name__ == "__main__":
    main()
