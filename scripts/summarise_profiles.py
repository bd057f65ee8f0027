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
        if "filter_phase_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
    return vals


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
rec = {
    "kernel": "filter_phase_kernel<double,double>",
    "chans": 256,
    "samples": 10000000,
    "fetch_size_kb_raw": f_kb,
    "write_size_kb_raw": w_kb,
    "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B; MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
    "algorithmic_bytes": 16 * 256 * 10000000,
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/collect_profiles.sh {tag})",
}
for name in (f"{tag}_filter_pmc_hbm.json", "hbm_traffic.json"):
    json.dump(rec, open(os.path.join(dst, name), "w"), indent=1)
print(json.dumps(rec, indent=1))
for r in ours[:12]:
    print(f"{r[0][:90]:<92} calls {r[1]:>5}  avg {float(r[3]) / 1e3:>10.1f} us  {r[4]:>6} %")
