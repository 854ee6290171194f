"""One-off phase timing of the CNN evaluator kernel (build with -DAR_CNN_PROF into libalpharat_hip_stats.so)."""
import sys
from pathlib import Path
sys.path.insert(0, ".")
from alpharat_amd import _lib
_lib.LIB_PATH = _lib.PKG / "libalpharat_hip_stats.so"
from alpharat_amd.sampling import rust_self_play
rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=2048, simulations=64, batch_size=16, output_dir=None,
               seed=0, concurrent_games=2048, weights_path=str(Path("tests/golden/nets/cnn_gpool_7x7_c64.arnet")))
