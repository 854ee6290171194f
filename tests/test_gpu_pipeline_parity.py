"""-m gpu: the NETWORK self-play pipeline -- k_gather -> device-wide leaf queue -> evaluator kernel ->
k_backup -> side-stream k_advance -> refill, the path bench.py times -- against the CPU oracle at record level.

The oracle's search (oracle/mcts.hpp, a restatement of search.rs:961-1073 / selfplay.rs:515-598) gets a Backend
whose evaluate_batch calls the product's HIP evaluator through the C-ABI (ar_net_evaluate), so both sides see the
same network bits; whole-game records (visit counts, policies, values, priors, actions, counters) must then be
equal byte for byte. That holds only if the device pipeline hands every leaf to the right game in the right
order (the MuxBackend behaviour, mux.rs:170-289), and if a leaf's evaluation does not depend on which tile of
which launch it lands in.

Full simulation budgets of BASELINE configs 3, 4 and 5, more games than resident slots (refill), >= 64 resident
slots (leaf queue, side-stream tree reuse all live). Bar: bit-exact."""
import ctypes as C
import time
from pathlib import Path

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden" / "nets"

TUNED = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)   # configs/mcts/7x7_rust_tuned.yaml
STRONG = dict(c_puct=0.512, fpu_reduction=0.479, force_k=0.025, noise_epsilon=0.25)  # configs/mcts/7x7_rust_strong.yaml


class HipEvaluator:
    """oracle Backend (kind 4) -> ar_net_evaluate: leaves become ArGameSpecs on the given maze."""

    def __init__(self, blob, w, h, max_turns, cost=None):
        from alpharat_amd import _lib
        from alpharat_amd.game import PyRat
        from alpharat_amd.nets import Net

        self._lib = _lib
        self.net = Net(blob)
        self.w, self.h, self.max_turns = w, h, max_turns
        self.cost = np.ascontiguousarray(PyRat.open_cost(w, h) if cost is None else cost, dtype=np.uint8).reshape(-1)
        self.backend = O.CallbackBackend(self.evaluate)

    def evaluate(self, leaves):
        _lib = self._lib
        n = len(leaves)
        hw = self.w * self.h
        cheese = [np.ascontiguousarray(l["cheese"][:hw], dtype=np.uint8) for l in leaves]
        specs = (_lib.ArGameSpec * n)(*[
            _lib.ArGameSpec(self.w, self.h, self.max_turns, l["turn"], l["p1"][0], l["p1"][1], l["p2"][0], l["p2"][1],
                            l["p1_mud"], l["p2_mud"], l["p1_score"], l["p2_score"],
                            self.cost.ctypes.data_as(C.c_void_p), cheese[i].ctypes.data_as(C.c_void_p))
            for i, l in enumerate(leaves)])
        p1 = np.zeros((n, 5), np.float32)
        p2 = np.zeros((n, 5), np.float32)
        v1 = np.zeros(n, np.float32)
        v2 = np.zeros(n, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(_lib.load().ar_net_evaluate(self.net.handle, specs, n, p(p1), p(p2), p(v1), p(v2), None, None))
        return p1, p2, v1, v2


def _check_game(g, want):
    from test_gpu_parity import _check_game as check

    check(g, want)


def _note(line):
    """where the time of these tests goes (kept next to the other run artefacts; never fails a test)"""
    try:
        d = Path(__file__).resolve().parent.parent / "gpurun_out"
        d.mkdir(exist_ok=True)
        with open(d / "pipeline_parity_timing.txt", "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name,blob,sims,search,n_games,n_oracle", [
    ("config3_mlp", "mlp_7x7_h256", 1897, TUNED, 96, 3),             # 7x7, PyRatMLP h256, 7x7_rust_tuned
    ("config4_symmetric", "symmetric_7x7_h256", 2693, STRONG, 96, 2),  # SymmetricMLP h256, 7x7_rust_strong
    # CNN + global pooling c64, 4096 sims: a 50-turn game is 12 800 device steps of 64 games each, so this case plays
    # one generation (64 games in 64 slots; refills are exercised by the two cases above and the cache tests)
    ("config5_cnn", "cnn_gpool_7x7_c64", 4096, TUNED, 64, 2),
])
def test_network_selfplay_records_bit_exact(name, blob, sims, search, n_games, n_oracle):
    from alpharat_amd.sampling import rust_self_play

    resident = 64
    games = {}
    t0 = time.perf_counter()
    stats = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=n_games, simulations=sims,
                           batch_size=16, output_dir=None, seed=0, concurrent_games=resident,
                           weights_path=str(GOLD / f"{blob}.arnet"), on_game=lambda g: games.__setitem__(g["game_index"], g),
                           **search)
    assert stats.total_games == n_games and sorted(games) == list(range(n_games))
    assert stats.total_nn_evals > 0 and stats.steps > 0
    _note(f"{name}: device run {time.perf_counter() - t0:.1f} s, {stats.steps} steps, {stats.total_nn_evals} evals, "
          f"{stats.total_positions} positions")
    ev = HipEvaluator(GOLD / f"{blob}.arnet", 7, 7, 50)
    cfg = O.make_config(**search)
    # the last game (it started in a refilled slot when there are more games than slots), the first, and one more
    for i in [n_games - 1, 0, 70][:n_oracle]:
        t0 = time.perf_counter()
        want = O.play_game(O.Game(7, 7, 50).random_cheese(10, True, i), cfg, sims, 16, 0xA1FA0000 + i, backend=4,
                           net=ev.backend, game_index=i)
        _note(f"{name}: oracle game {i}: {time.perf_counter() - t0:.1f} s, {want['n']} positions, {ev.backend.calls} calls so far")
        _check_game(games[i], want)
        assert want["total_nn_evals"] > 0 and max(ev.backend.sizes) <= 16


@pytest.mark.parametrize("shape", ["wide", "octet3", "lane"])
def test_network_selfplay_other_gather_shapes_bit_exact_vs_oracle(shape, monkeypatch):
    """The gather kernels with four, two and one lane per game (AR_GATHER; eight lanes is the default the cases above run
    on) against the oracle at record level: tuned constants + noise, 600 simulations, more games than slots. (That all
    shapes produce the same records as each other is test_selfplay_records_do_not_depend_on_scheduling.)"""
    from alpharat_amd.sampling import rust_self_play

    monkeypatch.setenv("AR_GATHER", shape)
    games = {}
    stats = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=40, simulations=600, batch_size=16,
                           output_dir=None, seed=4, concurrent_games=32, weights_path=str(GOLD / "mlp_7x7_h256.arnet"),
                           on_game=lambda g: games.__setitem__(g["game_index"], g), **TUNED)
    monkeypatch.delenv("AR_GATHER")
    assert stats.total_games == 40 and stats.total_nn_evals > 0
    ev = HipEvaluator(GOLD / "mlp_7x7_h256.arnet", 7, 7, 50)
    cfg = O.make_config(**TUNED)
    for i in (0, 39):
        want = O.play_game(O.Game(7, 7, 50).random_cheese(10, True, 4 + i), cfg, 600, 16, 0xA1FA0000 + 4 + i, backend=4,
                           net=ev.backend, game_index=i)
        _check_game(games[i], want)


def test_network_selfplay_with_eval_cache_bit_exact_vs_oracle():
    """cache_size > 0 (k_cache_probe / k_cache_fill between the gather and the network): hits must return the bits
    the network would compute again, so records still equal the oracle's (which evaluates every leaf)."""
    from alpharat_amd.sampling import rust_self_play

    games = {}
    stats = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=80, simulations=600, batch_size=16,
                           output_dir=None, seed=2, concurrent_games=64, cache_size=4096, num_threads=4,
                           weights_path=str(GOLD / "mlp_7x7_h256.arnet"),
                           on_game=lambda g: games.__setitem__(g["game_index"], g), **TUNED)
    assert stats.cache_hits > 0 and stats.cache_misses > 0
    ev = HipEvaluator(GOLD / "mlp_7x7_h256.arnet", 7, 7, 50)
    cfg = O.make_config(**TUNED)
    for i in (1, 66, 79):
        want = O.play_game(O.Game(7, 7, 50).random_cheese(10, True, 2 + i), cfg, 600, 16, 0xA1FA0000 + 2 + i, backend=4,
                           net=ev.backend, game_index=i)
        _check_game(games[i], want)


def test_eval_cache_keeps_generated_mazes_apart():
    """One maze per game living in a reused slot entry: a cached evaluation of an earlier game's maze must never
    answer for a later game in the same slot (the reference hashes walls and mud, cached_backend.rs:130-199).
    Small board, few resident slots, many games -> slots are reused often and positions repeat."""
    from alpharat_amd.sampling import rust_self_play

    def run(cache):
        games = {}
        st = rust_self_play(width=5, height=5, cheese_count=4, max_turns=20, num_games=60, simulations=96, batch_size=8,
                            output_dir=None, seed=9, concurrent_games=6, maze_type="random", wall_density=0.5,
                            mud_density=0.2, cache_size=cache, weights_path=str(GOLD / "mlp_5x5_h32.arnet"),
                            on_game=lambda g: games.__setitem__(g["game_index"], g))
        return st, games

    st_c, with_cache = run(1 << 14)
    _, without = run(0)
    assert st_c.cache_hits > 0
    for i, g in without.items():
        for k, v in g.items():
            if isinstance(v, np.ndarray):
                np.testing.assert_array_equal(v, with_cache[i][k], err_msg=f"game {i} {k}")
            else:
                assert v == with_cache[i][k], (i, k)
    # and two of them against the oracle driven by the HIP evaluator on that game's own maze
    cfg = O.make_config()
    for i in (7, 55):
        og = O.Game(5, 5, 20).random_maze(0.5, 0.2, True, 9 + i).random_cheese(4, True, 9 + i)
        ev = HipEvaluator(GOLD / "mlp_5x5_h32.arnet", 5, 5, 20, cost=og.cost())
        want = O.play_game(og, cfg, 96, 8, 0xA1FA0000 + 9 + i, backend=4, net=ev.backend, game_index=i)
        _check_game(with_cache[i], want)


def test_one_net_on_two_mazes_rebinds_its_maze_constants():
    """A persistent Net searched on maze A, then on maze B of the same size: the first-layer maze constants are
    those of the maze at hand (a stale binding would silently evaluate B's leaves with A's walls)."""
    from alpharat_amd.game import PyRat
    from alpharat_amd.mcts import rust_mcts_search
    from alpharat_amd.nets import Net

    blob = GOLD / "mlp_5x5_h32.arnet"
    cheese = [(2, 2), (1, 3), (3, 1), (0, 4)]
    a = PyRat.create_custom(5, 5, walls=[((0, 0), (0, 1)), ((2, 2), (3, 2))], cheese=cheese, max_turns=30)
    b = PyRat.create_custom(5, 5, walls=[((0, 0), (1, 0))], mud=[((2, 2), (2, 3), 3)], cheese=cheese, max_turns=30)
    shared = Net(blob)
    got = [rust_mcts_search(g, simulations=80, batch_size=8, seed=5, net=shared) for g in (a, b, a)]
    for g, r in zip((a, b, a), got):
        fresh = rust_mcts_search(g, simulations=80, batch_size=8, seed=5, net=Net(blob))
        for k in ("policy_p1", "policy_p2", "prior_p1", "prior_p2", "visit_counts_p1", "visit_counts_p2"):
            assert getattr(r, k).tobytes() == getattr(fresh, k).tobytes(), k
        assert (r.value_p1, r.value_p2, r.total_visits) == (fresh.value_p1, fresh.value_p2, fresh.total_visits)
    assert got[0].prior_p1.tobytes() != got[1].prior_p1.tobytes()  # the two mazes do differ at the root


def test_network_of_another_board_size_is_refused():
    from alpharat_amd.game import PyRat
    from alpharat_amd.mcts import rust_mcts_search
    from alpharat_amd.nets import Net
    from alpharat_amd.sampling import rust_self_play

    with pytest.raises(ValueError, match="board"):
        rust_self_play(width=10, height=10, cheese_count=6, max_turns=30, num_games=2, simulations=16, output_dir=None,
                       weights_path=str(GOLD / "mlp_5x5_h32.arnet"))
    with pytest.raises(ValueError, match="board"):
        rust_self_play(width=5, height=7, cheese_count=6, max_turns=30, num_games=2, simulations=16, output_dir=None,
                       weights_path=str(GOLD / "cnn_gpool_7x5_c16.arnet"))
    with pytest.raises(ValueError, match="board size"):
        rust_mcts_search(PyRat.create_custom(7, 7, cheese=[(3, 3)], max_turns=30), simulations=16, net=Net(GOLD / "mlp_5x5_h32.arnet"))


# ---- sessions: the same run in slices --------------------------------------------------------------------------
def test_session_slices_play_the_same_games_and_windows_add_up():
    from alpharat_amd.sampling import SelfPlaySession, rust_self_play

    kw = dict(width=7, height=7, cheese_count=10, max_turns=50, num_games=70, simulations=200, batch_size=16,
              output_dir=None, seed=3, concurrent_games=32, weights_path=str(GOLD / "mlp_7x7_h256.arnet"), **TUNED)
    whole = {}
    st = rust_self_play(on_game=lambda g: whole.__setitem__(g["game_index"], g), **kw)
    sliced = {}
    windows = []
    with SelfPlaySession(on_game=lambda g: sliced.__setitem__(g["game_index"], g), **kw) as s:
        while not s.finished:
            windows.append(s.step(37))  # not a multiple of the host-visit period
            assert len(windows) < 10000
        total = s.close()
    assert sorted(sliced) == sorted(whole) == list(range(70))
    for i, g in whole.items():
        for k, v in g.items():
            if isinstance(v, np.ndarray):
                np.testing.assert_array_equal(v, sliced[i][k], err_msg=f"game {i} {k}")
            else:
                assert v == sliced[i][k], (i, k)
    for k in ("total_games", "total_positions", "total_simulations", "total_nn_evals", "total_terminals",
              "total_collisions", "gather_node_visits", "backup_node_visits", "new_nodes", "p1_wins", "p2_wins", "draws"):
        assert sum(getattr(w, k) for w in windows) == getattr(st, k) == getattr(total, k), k
    assert sum(w.steps for w in windows) == total.steps
    assert all(w.steps <= 37 for w in windows) and windows[0].steps == 37
    # work is reported per finished move, not per finished game: the first window already has simulations
    assert windows[0].total_simulations > 0 and windows[0].total_positions > 0 and windows[0].total_games == 0


def test_unbounded_session_keeps_every_slot_busy():
    from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession

    seen = []
    with SelfPlaySession(width=5, height=5, cheese_count=5, max_turns=30, num_games=UNBOUNDED, simulations=64,
                         batch_size=8, output_dir=None, seed=1, concurrent_games=48, first_game_index=1000,
                         on_game=lambda g: seen.append(g["game_index"])) as s:
        ws = [s.step(200) for _ in range(4)]
        assert not s.finished
    assert all(w.total_simulations > 0 and w.steps == 200 for w in ws)
    assert sum(w.total_games for w in ws) == len(seen) > 48  # slots were refilled
    assert sorted(seen) == sorted(set(seen)) and min(seen) == 1000
    # games are handed out in index order: a finished game's index is below first + finished + resident
    assert max(seen) < 1000 + len(seen) + 48
