"""Wall clock of the pair gather's round loop per wavefront, from the instrumented build
(hipcc ... -DAR_STATS -o alpharat_amd/libalpharat_hip_stats.so; AR_GATHER=pair). Bench workload.
Usage: AR_GATHER=pair python tools/pair_clk.py [resident] [warm batch steps] [window batch steps]"""
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
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
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
    L.ar_debug_gather_clk(clk)
c = list(clk)
waves, rounds, ticks, longest, pre = c[0], c[1], c[2], c[3], c[4]
print(f"window: {st.steps} batch steps, device {st.device_secs:.3f} s, gather launches {st.gather_launches}, "
      f"mean launch {st.gather_secs / max(st.gather_launches, 1) * 1e3:.3f} ms")
print(f"wavefronts {waves}, mean rounds {rounds / max(waves, 1):.1f}, mean loop {ticks / max(waves, 1) / 100:.1f} us, "
      f"longest loop {longest / 100:.1f} us, per round {ticks / max(rounds, 1) / 100:.2f} us, entry->loop {pre / max(waves, 1) / 100:.1f} us")
for b in range(32):
    if c[8 + b]:
        us = c[40 + b] / c[8 + b] / 100
        print(f"   rounds {b * 16:>3}-{b * 16 + 15:<3}: {c[8 + b]:>9} wavefronts, mean loop {us:8.1f} us, {us / (b * 16 + 8):6.2f} us per round")
