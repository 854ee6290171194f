"""CPU: the C-ABI library loads and exports every symbol include/alpharat_hip.h declares; the
host-side Python mirrors keep the reference's names and signatures. No compute calls here."""
import ctypes as C
import inspect
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from alpharat_amd import _lib

    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from alpharat_amd import _lib

    header = (ROOT / "include" / "alpharat_hip.h").read_text()
    declared = set(re.findall(r"\b(ar_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_version_and_error_channel(lib):
    assert b"alpharat_hip" in lib.ar_version()
    buf = C.create_string_buffer(64)
    lib.ar_last_error(buf, 64)  # callable before any error


def test_no_device_fails_loudly_not_silently(lib):
    """In the GPU-less build container every compute entry point must return an error, never a result."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from alpharat_amd.game import PyRat
    from alpharat_amd.mcts import rust_mcts_search
    from alpharat_amd.sampling import rust_self_play

    with pytest.raises(RuntimeError, match="no HIP device|hipGetDeviceCount"):
        rust_mcts_search(PyRat.create_custom(5, 5, cheese=[(2, 2)]), simulations=10)
    with pytest.raises(RuntimeError, match="no HIP device|hipGetDeviceCount"):
        rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=1, simulations=10, output_dir=None)


def test_python_mirror_signatures_match_reference():
    from alpharat_amd.mcts import rust_mcts_search
    from alpharat_amd.sampling import SelfPlayProgress, SelfPlayStats, rust_self_play

    # crates/alpharat-mcts/src/bindings.rs:229 pyo3 signature
    p = inspect.signature(rust_mcts_search).parameters
    for name, default in dict(predict_fn=None, simulations=100, batch_size=8, c_puct=1.5, fpu_reduction=0.2, force_k=2.0,
                              noise_epsilon=0.0, noise_concentration=10.83, collision_limit_min=1, collision_limit_max=256,
                              collision_scaling_start=800, collision_scaling_end=50000, collision_scaling_power=1.0,
                              seed=None).items():
        assert p[name].default == default and p[name].kind is inspect.Parameter.KEYWORD_ONLY, name
    # crates/alpharat-sampling/src/bindings.rs:269-302 pyo3 signature
    p = inspect.signature(rust_self_play).parameters
    for name, default in dict(cheese_symmetric=True, maze_type="open", positions="corners", wall_density=0.7,
                              mud_density=0.1, maze_symmetric=True, batch_size=8, c_puct=1.5, fpu_reduction=0.2,
                              force_k=2.0, noise_epsilon=0.0, noise_concentration=10.83, num_threads=4,
                              max_games_per_bundle=32, onnx_model_path=None, device="auto", mux_max_batch_size=256,
                              cache_size=0, progress=None).items():
        assert p[name].default == default, name
    for name in ("width", "height", "cheese_count", "max_turns", "num_games", "simulations", "output_dir"):
        assert p[name].default is inspect.Parameter.empty
    s = SelfPlayStats(total_games=3, total_positions=30, total_simulations=300, elapsed_secs=2.0, total_nn_evals=100,
                      total_terminals=150, total_collisions=50, draws=1)
    assert "SelfPlayStats(" in repr(s) and "games=3" in repr(s)  # test_rust_sampling.py:92-107
    assert s.games_per_second == 1.5 and s.avg_turns == 10 and s.collision_fraction == 50 / 300
    assert s.nn_eval_fraction == 100 / 300 and s.draw_rate == 1 / 3
    pr = SelfPlayProgress()
    assert (pr.games_completed, pr.positions_completed, pr.simulations_completed, pr.nn_evals_completed) == (0, 0, 0, 0)


def test_shims_resolve_reference_import_names():
    import sys

    sys.path.insert(0, str(ROOT / "alpharat_amd" / "shims"))
    try:
        import alpharat_mcts
        import alpharat_sampling

        assert alpharat_mcts.rust_mcts_search and alpharat_mcts.SearchResult
        for n in ("rust_self_play", "SelfPlayStats", "SelfPlayProgress", "preload_cuda_libs", "preload_tensorrt_libs"):
            assert hasattr(alpharat_sampling, n)
        alpharat_sampling.preload_cuda_libs()
    finally:
        sys.path.pop(0)


def test_python_game_matches_oracle_engine():
    """The PyRat-compatible Python class follows the same rules as the oracle engine."""
    import numpy as np

    import _oracle as O
    from alpharat_amd.game import PyRat

    rng = np.random.default_rng(0)
    og = O.Game(5, 5, 40, p1=(2, 2), p2=(4, 4), cheese=[(0, 0), (4, 0), (2, 3), (1, 1)], mud=[((2, 2), (2, 3), 3)],
                walls=[((0, 0), (0, 1)), ((3, 3), (4, 3))])
    pg = PyRat.create_custom(5, 5, walls=[((0, 0), (0, 1)), ((3, 3), (4, 3))], mud=[((2, 2), (2, 3), 3)],
                             cheese=[(0, 0), (4, 0), (2, 3), (1, 1)], player1_pos=(2, 2), player2_pos=(4, 4), max_turns=40)
    for _ in range(40):
        st = og.state()
        assert (tuple(pg.player1_position), tuple(pg.player2_position)) == (st["p1"], st["p2"])
        assert (pg.player1_score, pg.player2_score, pg.player1_mud_turns, pg.turn) == (
            st["p1_score"], st["p2_score"], st["p1_mud"], st["turn"])
        e1, e2 = og.effective_actions()
        assert pg.effective_actions_p1() == e1 and pg.effective_actions_p2() == e2
        assert pg.is_over() == og.over()
        if og.over():
            break
        a, b = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        og.make_move(a, b)
        pg.make_move(a, b)
