"""15x11 board (configs/game/15x11_open_asymmetric.yaml shape: > 64 cells, 4 cheese words), uniform evaluator."""
import sys, time
sys.path.insert(0, ".")
from alpharat_amd.sampling import rust_self_play
for n in (8192, 8192):
    t0 = time.perf_counter()
    st = rust_self_play(width=15, height=11, cheese_count=21, max_turns=150, num_games=n, simulations=400, batch_size=16,
                        output_dir=None, seed=0, concurrent_games=n, cheese_symmetric=False)
    dt = time.perf_counter() - t0
    print(f"15x11 uniform {n} games: {dt:.2f}s wall, {st.device_secs:.2f}s device, {st.total_simulations / dt / 1e6:.1f}M sims/s, steps {st.steps}", flush=True)
