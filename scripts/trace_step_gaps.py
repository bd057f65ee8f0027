#!/usr/bin/env python3
"""Idle time of the device inside one bench step, from a rocprofv3 kernel trace of bench.py (collect_profiles.sh):
kernel time, gaps by (kernel before, kernel after).   python scripts/trace_step_gaps.py gpurun_out/profiles_<tag>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/bench/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "absdiff_partial" in r["Kernel_Name"]]
i0, i1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
step = rows[i0:i1]
wall = (int(rows[i1]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6
print(f"one step: wall {wall:.2f} ms, kernels {busy:.2f} ms in {len(step)} launches, idle {wall - busy:.2f} ms")
gaps = collections.Counter()
count = collections.Counter()
for a, b in zip(step[:-1], step[1:]):
    k = (a["Kernel_Name"].split("(")[-2][-40:] if False else a["Kernel_Name"][:44], b["Kernel_Name"][:44])
    gaps[k] += int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    count[k] += 1
for k, v in gaps.most_common(8):
    print(f"  {v / 1e6:6.2f} ms in {count[k]:3d} gaps ({v / count[k] / 1e3:6.1f} us each)  {k[0]}  ->  {k[1]}")
