#!/usr/bin/env python3
"""profiles/traffic.json: HBM bytes per launch of the tree kernels, from two rocprofv3 PMC passes
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs as MI355X_MICROARCH.md prescribes: TCC has four counter
slots, FETCH_SIZE takes three and WRITE_SIZE two) of the SAME bench command, keyed by a hash of the kernel sources
so that bench.py reports the figure only for the build it was measured on.

    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR --evaluator mlp --resident 65536 --source "profiles/r02_pmc_hbm_traffic.txt"

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are KB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so it is doubled (calibrated for wide streaming reads; these kernels issue scattered
16-byte reads, for which the factor is an upper bound -- the JSON keeps the raw figure beside it); WRITE_SIZE is
taken as read.
"""
from __future__ import annotations

import argparse
import csv
import json
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def per_dispatch(d: str, counter: str) -> dict[str, tuple[float, int]]:
    tot: dict[str, float] = defaultdict(float)
    ids: dict[str, set] = defaultdict(set)

    def short(name: str) -> str:
        return name.replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]

    for f in Path(d).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    k = short(row["Kernel_Name"])
                    tot[k] += float(row["Counter_Value"])
                    ids[k].add(row["Dispatch_Id"])
    for f in Path(d).rglob("*.db"):
        c = sqlite3.connect(f)
        try:
            cur = c.execute("select * from counters_collection")
        except sqlite3.Error:
            continue
        cols = [x[0] for x in cur.description]
        ik, ic, iv, idp = (cols.index(n) for n in ("kernel_name", "counter_name", "value", "dispatch_id"))
        for row in cur:
            if row[ic] == counter:
                k = short(row[ik])
                tot[k] += float(row[iv])
                ids[k].add(row[idp])
    return {k: (tot[k] / max(len(ids[k]), 1), len(ids[k])) for k in tot}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--evaluator", default="mlp")
    ap.add_argument("--resident", type=int, default=65536)
    ap.add_argument("--source", default="")
    ap.add_argument("--out", default=str(ROOT / "profiles" / "traffic.json"))
    a = ap.parse_args()
    import bench

    fetch = per_dispatch(a.fetch_dir, "FETCH_SIZE")
    write = per_dispatch(a.write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kb, nf = fetch.get(k, (0.0, 0))
        w_kb, nw = write.get(k, (0.0, 0))
        kernels[k] = {
            "fetch_kb_per_launch_raw": f_kb, "write_kb_per_launch": w_kb, "launches": [nf, nw],
            "bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
            "bytes_per_launch_uncorrected": (f_kb + w_kb) * 1024.0,
        }
    out = {"source_hash": bench.kernel_source_hash(), "evaluator": a.evaluator, "resident": a.resident,
           "source": a.source, "correction": "FETCH_SIZE x2 (gfx950 wide-read tally) + WRITE_SIZE, KB -> bytes",
           "kernels": kernels}
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    for k, v in kernels.items():
        print(f"{k:<24} fetch {v['fetch_kb_per_launch_raw'] / 1024:9.1f} MB (raw)  write {v['write_kb_per_launch'] / 1024:9.1f} MB  "
              f"-> {v['bytes_per_launch'] / 1e6:9.1f} MB per launch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
