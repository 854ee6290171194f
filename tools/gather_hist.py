"""Distribution of gather rounds per lane / per wavefront and of the paths lanes take, from the
instrumented build (hipcc ... -DAR_STATS -o alpharat_amd/libalpharat_hip_stats.so). Bench workload."""
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
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3072   # batch steps before the measured window (steady state)
window = int(sys.argv[3]) if len(sys.argv) > 3 else 512
blob = bench.make_mlp_blob(ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet")
search, sims, batch, _ = bench.WORKLOADS["mlp"]
L.ar_debug_gather_hist.argtypes = [C.c_void_p]
L.ar_debug_gather_clk.argtypes = [C.c_void_p]
out = (C.c_ulonglong * 136)()
clk = (C.c_ulonglong * 128)()
with SelfPlaySession(**bench.GAME, num_games=UNBOUNDED, simulations=sims, batch_size=batch, output_dir=None,
                     weights_path=str(blob), seed=0, concurrent_games=resident, **search) as s:
    s.step(warm)
    L.ar_debug_gather_hist(out)  # (reading resets the counters)
    L.ar_debug_gather_clk(clk)
    st = s.step(window)
L.ar_debug_gather_hist(out)
o = list(out)
lanes, waves, paths = o[:64], o[64:128], o[128:136]


def show(name, h):
    tot = sum(h)
    mean = sum((i * 4 + 2) * c for i, c in enumerate(h)) / max(tot, 1)
    print(f"{name}: n={tot} mean rounds ~{mean:.1f}")
    acc = 0
    for i, c in enumerate(h):
        acc += c
        if c:
            print(f"   {i * 4:>3}-{i * 4 + 3:<3} {100.0 * c / tot:6.2f}%  cum {100.0 * acc / tot:6.2f}%")


show("lanes (complete gathers)", lanes)
show("wavefronts (max over lanes)", waves)
names = ["pop level", "root pick", "new leaf", "leaf claim/collision", "interior (expand)", "allocation steps",
         "rounds spent only allocating", "finish at pick"]
tot = sum(paths[:5]) + paths[6] + paths[7]
print("lane-rounds by path:")
for n, c in zip(names, paths):
    print(f"   {n:<30} {c:>14}  {100.0 * c / max(tot, 1):6.2f}% of lane-rounds")
L.ar_debug_gather_clk(clk)
clk = list(clk)
print("gather loop wall clock (100 MHz ticks -> us) by wavefront max rounds:")
for b in range(48):
    if clk[64 + b]:
        us = clk[b] / clk[64 + b] / 100.0
        print(f"   rounds {b * 8:>3}-{b * 8 + 7:<3} waves {clk[64 + b]:>9}  mean {us:9.1f} us  per round {us / (b * 8 + 4):6.2f} us")
print("mean time since the loop began after N wavefront rounds:")
for cp in range(8):
    if clk[120 + cp]:
        print(f"   after {1 << cp:>3} rounds: {clk[112 + cp] / clk[120 + cp] / 100.0:9.1f} us  ({clk[120 + cp]} wavefronts)")
print("steps", st.steps, "device_secs", st.device_secs)
