import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from pathlib import Path
from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession
GOLD = Path('/root/repo/tests/golden/nets')
TUNED = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
res = int(sys.argv[1]); blob = sys.argv[2]
n=[0]
with SelfPlaySession(width=7, height=7, cheese_count=10, max_turns=50, num_games=UNBOUNDED, simulations=1897, batch_size=16, output_dir=None,
                     weights_path=blob, seed=0, concurrent_games=res, **TUNED) as s:
    print(s.info(), flush=True)
    for k in range(int(sys.argv[3])):
        t=time.perf_counter(); w=s.step(1024); dt=time.perf_counter()-t
        print(k, f"{dt:.2f}s sims/s {w.total_simulations/dt/1e6:.1f}M games {w.total_games} nn/s {w.total_nn_evals/dt/1e6:.1f}M", s.info()['host_grown_arenas'], flush=True)
