"""alpharat_amd -- MI355X-native self-play MCTS sampler for simultaneous-move PyRat.

Only the hot path of mintiti/alpharat lives here: decoupled-PUCT tree search, the PyRat step,
the policy/value heads and game recording, as HIP kernels behind a C-ABI (include/alpharat_hip.h,
libalpharat_hip.so). The Python in this package mirrors the reference's Python-facing interface
for that path (alpharat_mcts.rust_mcts_search, alpharat_sampling.rust_self_play, ...).
"""

__version__ = "0.1.0"
