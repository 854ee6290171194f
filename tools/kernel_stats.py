#!/usr/bin/env python3
"""Per-kernel launch statistics from a `rocprofv3 --kernel-trace --output-format csv -d DIR` run: reads every
*kernel_trace.csv under DIR and prints name,calls,total_us,avg_us,pct (sorted by total).
Usage: python tools/kernel_stats.py DIR > profiles/rNN_..._kernel_stats.csv"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

tot = defaultdict(float)
cnt = defaultdict(int)
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            tot[name] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            cnt[name] += 1
total = sum(tot.values()) or 1.0
print("name,calls,total_us,avg_us,pct")
for name in sorted(tot, key=lambda n: -tot[n]):
    print(f"\"{name}\",{cnt[name]},{tot[name]:.1f},{tot[name] / cnt[name]:.2f},{100 * tot[name] / total:.2f}")
