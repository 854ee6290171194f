"""Quick throughput probe of the self-play path (not the benchmark)."""
import sys, time, json
sys.path.insert(0, ".")
from alpharat_amd.sampling import rust_self_play

def run(w, h, cheese, turns, games, conc, sims, batch, **kw):
    t = time.time()
    s = rust_self_play(width=w, height=h, cheese_count=cheese, max_turns=turns, num_games=games, simulations=sims,
                       batch_size=batch, output_dir=None, seed=0, concurrent_games=conc, **kw)
    nv = s.gather_node_visits + s.backup_node_visits
    print(json.dumps(dict(cfg=f"{w}x{h} sims={sims} games={games} conc={conc}", wall=round(time.time() - t, 3),
                          elapsed=round(s.elapsed_secs, 3), device=round(s.device_secs, 3), steps=s.steps,
                          games_s=round(s.games_per_second, 1), sims_s=round(s.simulations_per_second),
                          descents_s=round((s.total_nn_evals + s.total_terminals) / s.elapsed_secs),
                          node_visits=nv, gbs_dev=round(nv * 300 / max(s.device_secs, 1e-9) / 1e9, 2),
                          positions=s.total_positions)), flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        run(5, 5, 5, 30, 256, 256, 200, 8)
        run(5, 5, 5, 30, 1024, 1024, 1000, 16)
        run(7, 7, 10, 50, 1024, 1024, 400, 16, c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
    else:
        run(5, 5, 5, 30, 4096, 4096, 1000, 16)
        run(7, 7, 10, 50, 8192, 8192, 1897, 16, c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
