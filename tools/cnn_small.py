"""CNN self-play with few resident games (the parity test's regime): where does a step's time go?"""
import sys, time
sys.path.insert(0, ".")
from alpharat_amd.sampling import rust_self_play
t0 = time.perf_counter()
st = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=64, simulations=512, batch_size=16,
                    output_dir=None, seed=0, concurrent_games=64, weights_path="tests/golden/nets/cnn_gpool_7x7_c64.arnet",
                    c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
dt = time.perf_counter() - t0
print(f"{dt:.2f} s, {st.steps} steps, {dt / st.steps * 1e3:.3f} ms/step, {st.total_nn_evals} evals")
