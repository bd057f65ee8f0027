#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (scripts/collect_profiles.sh) into the files profiles/ keeps:

  profiles/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
                                          (this library's kernels first; torch's synthetic-data kernels last)
  profiles/<tag>_filter_pmc_hbm.json      HBM bytes per filter launch from FETCH_SIZE / WRITE_SIZE (separate passes)
  profiles/hbm_traffic.json               the same, read by bench.py for roofline.traffic

    python scripts/summarise_profiles.py r02a
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(ROOT, "profiles")

stats = glob.glob(os.path.join(src, "bench", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.reader(open(stats)))
head, body = rows[0], rows[1:]
ours = [r for r in body if "parrm" in r[0] or "anonymous namespace" in r[0] and "at::native" not in r[0]]
main_kernel = "parrm_comb_kernel (generated per filter)" if any("parrm_comb_kernel" in r[0] for r in body) else "filter_phase_kernel<double,double>"
rest = [r for r in body if r not in ours]
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_ALL)
    w.writerow(head)
    w.writerows(ours + rest)
shutil.copy(os.path.join(src, "bench_stdout.log"), os.path.join(dst, f"{tag}_bench_stdout.log"))


def counter(kind, name):
    path = glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = []
    for r in csv.DictReader(open(path)):
        if (r["Kernel_Name"].strip() == "parrm_comb_kernel" or "filter_phase_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
    return vals


def headline(vals):
    """The launches of the headline shape only (a plan's first use also runs a self-test launch of a few
    hundred thousand samples, which must not be averaged in)."""
    big = [v for v in vals if v > 0.5 * max(vals)]
    return sum(big) / len(big)


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
f_kb, w_kb = headline(fetch), headline(write)
rec = {
    "kernel": main_kernel,
    "chans": 256,
    "samples": 10000000,
    "fetch_size_kb_raw": f_kb,
    "write_size_kb_raw": w_kb,
    "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B; MI355X_MICROARCH.md, HBM section: calibrated for 16 B/lane streaming loads, which is what the generated kernel issues); WRITE_SIZE as reported",
    "bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
    "algorithmic_bytes": 16 * 256 * 10000000,
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/collect_profiles.sh {tag})",
}
for name in (f"{tag}_filter_pmc_hbm.json", "hbm_traffic.json"):
    json.dump(rec, open(os.path.join(dst, name), "w"), indent=1)
print(json.dumps(rec, indent=1))
# the filter kernel's launches of the headline shape in the bench trace (kernel_stats averages every launch of a
# name, and the generated kernel also runs a self-test and the configs[1] shape in the same process)
trace = glob.glob(os.path.join(src, "bench", "**", "*kernel_trace.csv"), recursive=True)[0]
durs = {}
for r in csv.DictReader(open(trace)):
    if "parrm_comb_kernel" in r["Kernel_Name"] or "filter_phase_kernel<double, double" in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0][:40], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
        durs.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(dst, f"{tag}_filter_kernel_launches.txt"), "w") as fh:
    fh.write("# filter kernel launches in the rocprofv3 kernel trace of the default bench command, by (kernel, grid size)\n")
    for (name, grid), v in sorted(durs.items(), key=lambda kv: -max(kv[1])):
        # one grid size can serve two shapes (256 workgroups: 256 ch x 10 M in one stretch per channel, and 64 ch x 1 M
        # in four): split by duration
        for label, part in (("long", [d for d in v if d > 0.5 * max(v)]), ("short", [d for d in v if d <= 0.5 * max(v)])):
            if part:
                line = (f"{name:<42} grid {grid:>9} {label:<5} launches {len(part):>3}  avg {sum(part) / len(part):8.3f} ms  "
                        f"min {min(part):8.3f}  max {max(part):8.3f}")
                fh.write(line + "\n")
                print(line)
for r in ours[:12]:
    print(f"{r[0][:90]:<92} calls {r[1]:>5}  avg {float(r[3]) / 1e3:>10.1f} us  {r[4]:>6} %")
