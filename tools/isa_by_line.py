#!/usr/bin/env python3
"""Static instruction counts of one kernel by source line, from the assembly hipcc leaves behind with
`-gline-tables-only --save-temps=obj` (the .loc directives). Usage: isa_by_line.py FILE.s KERNEL_SUBSTRING [top_n]

The gather kernels execute nearly all of a round's code in every round (eight games per wavefront, each in another
branch), so the static count of the round loop is close to the dynamic count per round."""
import collections
import re
import sys


def main() -> int:
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    files: dict[int, str] = {}
    by_line: dict[tuple[str, int], collections.Counter] = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    inside = False
    cur = ("?", 0)
    for raw in open(path, errors="replace"):
        line = raw.strip()
        m = re.match(r'\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if line.endswith(":") and key in line and not line.startswith("."):
                inside = True
            elif re.match(r"^[_A-Za-z0-9]+:", line) and key in line.split(":")[0]:
                inside = True
            continue
        if line.startswith(".Lfunc_end") or line.startswith(".amdhsa_kernel"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not line or line[0] in ".;" or line.endswith(":"):
            continue
        op = line.split()[0]
        kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        by_line[cur][kind] += 1
        total[kind] += 1
    print("total", dict(total))
    rows = sorted(by_line.items(), key=lambda kv: -kv[1]["valu"])[:top]
    for (f, ln), c in rows:
        print(f"{f}:{ln:<5} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
