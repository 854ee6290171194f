"""Self-play throughput with each evaluator family on the 7x7 golden networks (small run)."""
import sys, time
from pathlib import Path
sys.path.insert(0, ".")
from alpharat_amd.sampling import rust_self_play
G = Path("tests/golden/nets")
import os
GAMES, SIMS = int(os.environ.get("GAMES", "8192")), int(os.environ.get("SIMS", "400"))
for name in sys.argv[1:] or ["mlp_7x7_h256", "symmetric_7x7_h256", "cnn_gpool_7x7_c64"]:
    t0 = time.perf_counter()
    st = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=GAMES, simulations=SIMS, batch_size=16,
                        output_dir=None, seed=0, concurrent_games=GAMES, weights_path=str(G / f"{name}.arnet"),
                        c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
    dt = time.perf_counter() - t0
    print(f"{name:<22} {dt:6.2f}s wall {st.device_secs:6.2f}s device  {st.total_nn_evals / dt / 1e6:7.2f}M evals/s  "
          f"{st.total_simulations / dt / 1e6:7.1f}M sims/s  steps {st.steps}", flush=True)
