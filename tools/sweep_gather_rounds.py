"""Sweep the gather round limit (AR_GATHER_ROUNDS) on the bench workload, one process, arena kept
between runs. Usage: python tools/sweep_gather_rounds.py GAMES[/RESIDENT] R1[:LANES[:ALLOC_PER_ROUND]] ... ...   (0 = no limit)"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
from alpharat_amd.sampling import rust_self_play  # noqa: E402

games, _, conc = sys.argv[1].partition("/")
games = int(games)
conc = int(conc or games)
blob = bench.make_mlp_blob(ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet")
for r in sys.argv[2:]:
    r, _, rest = r.partition(":")
    lanes, _, apr = rest.partition(":")
    os.environ["AR_GATHER_ROUNDS"] = r
    os.environ["AR_LANES_PER_WAVE"] = lanes or "64"
    os.environ["AR_ALLOC_PER_ROUND"] = apr or "2"
    t0 = time.perf_counter()
    st = rust_self_play(**bench.GAME, num_games=games, simulations=bench.SIMS, batch_size=bench.BATCH, output_dir=None,
                        weights_path=str(blob), seed=0, first_game_index=0, concurrent_games=conc, **bench.SEARCH)
    dt = time.perf_counter() - t0
    print(f"rounds={r:>4} lanes={lanes or 64:>2} alloc/round={apr or 2:>2} wall={dt:7.2f}s device={st.device_secs:7.2f}s steps={st.steps} "
          f"sims/s={st.total_simulations / dt / 1e6:7.1f}M games/s={st.total_games / dt:7.1f} "
          f"avg_step_ms={st.device_secs / max(st.steps, 1) * 1e3:.3f}", flush=True)
