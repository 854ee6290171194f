"""Counters of the work-queue gather (k_gatherw) from the instrumented build
(hipcc ... -DAR_STATS -o alpharat_amd/libalpharat_hip_stats.so). Bench workload, steady state.
Usage: python tools/gw_stats.py [resident] [warm batch steps] [window batch steps]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from alpharat_amd import _lib  # noqa: E402

_lib.LIB_PATH = _lib.PKG / "libalpharat_hip_stats.so"
import bench  # noqa: E402
from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession  # noqa: E402

L = _lib.load()
resident = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
window = int(sys.argv[3]) if len(sys.argv) > 3 else 256
blob = bench.make_mlp_blob(ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet")
search, sims, batch, _ = bench.WORKLOADS["mlp"]
L.ar_debug_gather_clk.argtypes = [C.c_void_p]
clk = (C.c_ulonglong * 128)()
with SelfPlaySession(**bench.GAME, num_games=UNBOUNDED, simulations=sims, batch_size=batch, output_dir=None,
                     weights_path=str(blob), seed=0, concurrent_games=resident, **search) as s:
    s.step(warm)
    L.ar_debug_gather_clk(clk)  # (reading resets the counters)
    st = s.step(window)
    th = (C.c_ulonglong * 131)()
    L.ar_debug_tree_hist.argtypes = [C.c_void_p]
    L.ar_debug_tree_hist(th)
L.ar_debug_gather_clk(clk)
c = list(clk)
waves = max(c[0], 1)
names = ["passes", "items", "begin items", "pick ends", "waits (re-queued)", "interior entries", "final entries"]
print(f"launches {st.gather_launches}  avg launch {st.gather_secs / max(st.gather_launches, 1) * 1e3:.3f} ms  wavefronts {c[0]} ({c[0] / max(st.gather_launches, 1):.0f} per launch)")
for i, n in enumerate(names):
    print(f"  {n:<20} {c[1 + i]:>14}  per wavefront {c[1 + i] / waves:10.1f}  per launch {c[1 + i] / max(st.gather_launches, 1):12.0f}")
print(f"  items per pass {c[2] / max(c[1], 1):.1f}")
ph = ["pop + fetch (1)", "visit (2)", "publish (3)", "pick end / begin (4)", "whole loop"]
for i, n in enumerate(ph):
    print(f"  {n:<22} {c[8 + i] / waves / 100.0:10.1f} us per wavefront   {c[8 + i] / max(c[1], 1) / 100.0:8.2f} us per pass")
P = max(c[1], 1)
print(f"  allocation steps: slowest lane of a pass {c[40] / P:.2f}, all entries {c[41] / max(c[6], 1):.2f} per interior entry; root entries per pass {c[45] / P:.2f}")
print(f"  slowest lane of a pass, from the start of phase 2: record arrived {c[42] / P / 100.0:.2f} us, set-up done {c[43] / P / 100.0:.2f} us, "
      f"allocation done {c[44] / P / 100.0:.2f} us")
print("  items taken per pass (histogram by 4):")
tot = sum(c[16:33]) or 1
for b in range(17):
    if c[16 + b]:
        print(f"    {b * 4:>2}-{b * 4 + 3:<2} {100.0 * c[16 + b] / tot:6.2f}%")
print("  passes from a game's begin to its end (histogram by 8):")
tot = sum(c[64:112]) or 1
acc = 0
for b in range(48):
    acc += c[64 + b]
    if c[64 + b]:
        print(f"    {b * 8:>3}-{b * 8 + 7:<3} {100.0 * c[64 + b] / tot:6.2f}%  cum {100.0 * acc / tot:7.3f}%")
print("steps", st.steps, "device_secs", st.device_secs, "step ms", st.device_secs / max(st.steps, 1) * 1e3)
th = list(th)
n = max(th[130], 1)
print(f"resident trees: {th[130]} games, mean top-of-tree {th[128] / n:.0f} nodes, mean arena capacity {th[129] / n:.0f} nodes")
acc = 0
for b in range(64):
    if th[b] or th[64 + b]:
        acc += th[b]
        print(f"   {b * 512:>6}+  top {100.0 * th[b] / n:6.2f}% (cum {100.0 * acc / n:6.2f}%)   capacity {100.0 * th[64 + b] / n:6.2f}%")
