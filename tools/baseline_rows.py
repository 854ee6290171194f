#!/usr/bin/env python3
"""BASELINE.md section 4, rows C1 and C2, reproducibly:

    python tools/baseline_rows.py            # prints one JSON object per row

C1 (plumbing, CPU): 5x5 open, 5 cheese, 30 turns, default search (100 sims, batch 8, c_puct 1.5, force_k 2, fpu 0.2),
   64 games seeded 0..63 on the oracle's thread-per-game loop -- the configuration the reference's own sampler runs
   on a CPU (configs[0] of BASELINE.json); needs no GPU.
C2 (tree kernels only, GPU): the same game family, SmartUniform priors, 1000 sims, batch 16, 4096 concurrent games,
   game i seeded i, search stream 0xA1FA0000 + i.
"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def c1():
    import _oracle as O

    threads = min(os.cpu_count() or 1, 64)
    r = O.selfplay_bench(5, 5, 5, 30, 64, O.make_config(), 100, 8, threads)
    return {"row": "C1", "where": f"CPU oracle, {threads} threads", "games": r["games"], "games_per_sec": r["games"] / r["elapsed_secs"],
            "simulations_per_sec": r["simulations"] / r["elapsed_secs"], "positions": r["positions"],
            "node_visits_per_sec": (r["gather_node_visits"] + r["backup_node_visits"]) / r["elapsed_secs"]}


def c2():
    from alpharat_amd.sampling import rust_self_play

    kw = dict(width=5, height=5, cheese_count=5, max_turns=30, num_games=4096, simulations=1000, batch_size=16,
              output_dir=None, seed=0, concurrent_games=4096)
    rust_self_play(**kw)  # warm-up: arena allocation, first-touch
    t0 = time.perf_counter()
    st = rust_self_play(**kw)
    dt = time.perf_counter() - t0
    nv = st.gather_node_visits + st.backup_node_visits
    bytes_ = 300 * nv + 304 * st.new_nodes
    return {"row": "C2", "where": "1 MI355X, k_gather8 + k_uniform_eval + k_backup16 + k_advance (AR_UNIFORM=fused: k_step_uniform)", "games": st.total_games, "wall_s": dt,
            "games_per_sec": st.total_games / dt, "simulations_per_sec": st.total_simulations / dt,
            "descents_per_sec": (st.total_nn_evals + st.total_terminals) / dt, "node_visits_per_sec": nv / dt,
            "device_secs": st.device_secs, "algorithmic_GBps_in_kernels": bytes_ / max(st.device_secs, 1e-9) / 1e9,
            "frac_of_8TBps": bytes_ / max(st.device_secs, 1e-9) / 8e12}


if __name__ == "__main__":
    print(json.dumps(c1()))
    try:
        print(json.dumps(c2()))
    except Exception as e:  # noqa: BLE001 -- no GPU here
        print(json.dumps({"row": "C2", "skipped": str(e)}))
