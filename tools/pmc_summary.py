"""Sum rocprofv3 --pmc counter output per kernel. Reads every *counter_collection.csv (or rocpd
*.db) under the given directories and prints one table: kernel, dispatches, counter totals and
per-dispatch means.  Usage: python tools/pmc_summary.py DIR [DIR ...]"""
import csv
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name.split("(")[0][:48]


for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
    for f in Path(d).rglob("*.db"):
        c = sqlite3.connect(f)
        try:
            cur = c.execute("select * from counters_collection")
        except sqlite3.Error:
            continue
        cols = [x[0] for x in cur.description]
        ik, ic, iv, idp = (cols.index(n) for n in ("kernel_name", "counter_name", "value", "dispatch_id"))
        for row in cur:
            k = short(row[ik])
            tot[k][row[ic]] += float(row[iv])
            disp[k][row[ic]].add(row[idp])

for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    print(f"== {k}")
    for cn in sorted(tot[k]):
        n = max(len(disp[k][cn]), 1)
        print(f"  {cn:<26} total {tot[k][cn]:.5e}  dispatches {n:>6}  per-dispatch {tot[k][cn] / n:.5e}")
