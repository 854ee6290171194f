#!/usr/bin/env python3
"""A stretch of the kernel timeline from a `rocprofv3 --kernel-trace --output-format csv` run: start, end, duration and
hardware queue of every kernel in a window late in the run (which kernels of which group of games run side by side).
Usage: python tools/timeline.py DIR [window_ms] > profiles/rNN_timeline.txt"""
import csv
import sys
from pathlib import Path

rows = []
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("arnet::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:22], r["Queue_Id"]))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 16e6
t0 = rows[int(len(rows) * 0.8)][0]
print("start_ms   end_ms  dur_ms queue kernel")
for s, e, n, q in rows:
    if t0 <= s < t0 + win and not n.startswith("__amd_rocclr"):
        print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  q{q:>2}  {n}")
