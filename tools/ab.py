"""A/B runs of the bench workload in one process (arena kept between runs).
Usage: python tools/ab.py GAMES[/RESIDENT] REPEAT "ENV1=a,ENV2=b" "ENV3=c" ...   ("-" = no overrides)
Every config is run REPEAT times, interleaved, so drift of the box shows up as spread within a config."""
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
repeat = int(sys.argv[2])
configs = sys.argv[3:]
blob = bench.make_mlp_blob(ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet")
knobs = ("AR_MLP_MT", "AR_GATHER_ROUNDS", "AR_LANES_PER_WAVE", "AR_ALLOC_PER_ROUND", "AR_GROUPS", "AR_NO_ADVANCE_OVERLAP")
for rep in range(repeat):
    for cfg in configs:
        for k in knobs:
            os.environ.pop(k, None)
        if cfg != "-":
            for kv in cfg.split(","):
                k, _, v = kv.partition("=")
                os.environ[k] = v
        cache = int(os.environ.pop("CACHE", "0"))  # pseudo-knob: CACHE=n passes cache_size=n
        t0 = time.perf_counter()
        st = rust_self_play(**bench.GAME, cache_size=cache, num_games=games, simulations=bench.SIMS, batch_size=bench.BATCH, output_dir=None,
                            weights_path=str(blob), seed=0, first_game_index=0, concurrent_games=conc, **bench.SEARCH)
        dt = time.perf_counter() - t0
        print(f"[{rep}] {cfg:<44} wall={dt:7.2f}s device={st.device_secs:7.2f}s steps={st.steps} "
              f"hits={st.cache_hit_rate:5.3f} sims/s={st.total_simulations / dt / 1e6:7.1f}M avg_step_ms={st.device_secs / max(st.steps, 1) * 1e3:.3f}",
              flush=True)
