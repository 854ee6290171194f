"""-m gpu: the HIP path, called through the C-ABI, against the CPU oracle on the same seeded inputs.
Bar: bit-exact (visit counts, policies, values, selected actions, whole-game records)."""
from pathlib import Path

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

TUNED = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103)


def _pyrat(og: O.Game, max_turns):
    from alpharat_amd.game import PyRat

    st = og.state()
    maze = og.maze().reshape(-1).astype(np.int16)
    cost = np.where(maze < 0, 0, maze).astype(np.uint8)
    return PyRat(og.w, og.h, cost, og.cheese_mask(), st["p1"], st["p2"], max_turns, st["turn"], st["p1_score"],
                 st["p2_score"], st["p1_mud"], st["p2_mud"])


def _games():
    yield "open5_corner", O.Game(5, 5, 100, p1=(0, 0), p2=(4, 4), cheese=[(2, 2), (1, 3), (3, 1)]), 100
    yield "same_cell", O.Game(5, 5, 100, p1=(2, 2), p2=(2, 2), cheese=[(i, 0) for i in range(5)]), 100
    yield "short", O.Game(5, 5, 3, p1=(0, 0), p2=(2, 0), cheese=[(1, 0)]), 3
    g = O.Game(5, 5, 100, p1=(2, 2), p2=(4, 4), cheese=[(0, 0), (4, 0)], mud=[((2, 2), (2, 3), 3)],
               walls=[((0, 0), (0, 1)), ((3, 3), (4, 3))])
    g.make_move(0, 4)
    yield "mud_wall", g, 100
    yield "7x7", O.Game(7, 7, 50).random_cheese(10, True, 5), 50
    yield "15x11", O.Game(15, 11, 200).random_cheese(21, True, 2), 200
    t = O.Game(5, 5, 1, p1=(0, 0), p2=(0, 1), cheese=[(4, 4)])
    t.make_move(4, 4)
    yield "terminal_root", t, 1


def _assert_result(got, want, name):
    for k in ("policy_p1", "policy_p2", "visit_counts_p1", "visit_counts_p2", "prior_p1", "prior_p2"):
        assert getattr(got, k).tobytes() == want[k].tobytes(), (name, k, getattr(got, k), want[k])
    assert np.float32(got.value_p1).tobytes() == np.float32(want["value_p1"]).tobytes(), name
    assert np.float32(got.value_p2).tobytes() == np.float32(want["value_p2"]).tobytes(), name
    assert (got.total_visits, got.nn_evals, got.terminals, got.collisions) == (
        want["total_visits"], want["nn_evals"], want["terminals"], want["collisions"]), name


@pytest.mark.parametrize("sims,batch", [(1, 1), (50, 1), (200, 8), (1000, 16)])
def test_search_bit_exact(sims, batch):
    from alpharat_amd.mcts import rust_mcts_search

    for name, og, mt in _games():
        for kw in (dict(), TUNED, dict(noise_epsilon=0.25, **TUNED)):
            want = O.search_once(og, O.make_config(**kw), sims, batch, seed=42)
            got = rust_mcts_search(_pyrat(og, mt), simulations=sims, batch_size=batch, seed=42, **kw)
            _assert_result(got, want, (name, kw))


def test_search_many_matches_individual_searches():
    from alpharat_amd.mcts import search_many

    items = list(_games())[:5]
    res = search_many([_pyrat(og, mt) for _, og, mt in items], simulations=300, batch_size=8,
                      seeds=[7 + i for i in range(len(items))], **TUNED)
    for i, (name, og, mt) in enumerate(items):
        _assert_result(res[i], O.search_once(og, O.make_config(**TUNED), 300, 8, seed=7 + i), name)


def test_predict_fn_callback_path():
    # test_callback.py:114-133: batches never exceed batch_size; constant-value evaluator
    from alpharat_amd.mcts import rust_mcts_search

    og = O.Game(5, 5, 100, p1=(1, 1), p2=(3, 3), cheese=[(2, 2), (0, 4)])
    sizes = []

    def predict_fn(games):
        sizes.append(len(games))
        n = len(games)
        p1 = np.zeros((n, 5), np.float32)
        p2 = np.zeros((n, 5), np.float32)
        for i, g in enumerate(games):
            for arr, eff in ((p1, g.effective_actions_p1()), (p2, g.effective_actions_p2())):
                u = sorted(set(eff))
                for a in u:
                    arr[i, a] = np.float32(1.0) / np.float32(len(u))
        return p1, p2, np.full(n, 1.5, np.float32), np.full(n, 0.5, np.float32)

    want = O.search_once(og, O.make_config(), 120, 4, seed=123, backend=1, v1=1.5, v2=0.5)
    got = rust_mcts_search(_pyrat(og, 100), predict_fn=predict_fn, simulations=120, batch_size=4, seed=123)
    _assert_result(got, want, "callback")
    assert sizes and max(sizes) <= 4

    def boom(games):
        raise ValueError("nope")

    with pytest.raises(RuntimeError, match="predict_fn raised"):
        rust_mcts_search(_pyrat(og, 100), predict_fn=boom, simulations=20, batch_size=4, seed=1)


def _check_game(g, want):
    assert g["n"] == want["n"]
    np.testing.assert_array_equal(g["p1_pos"], want["ints"][:, 0:2])
    np.testing.assert_array_equal(g["p2_pos"], want["ints"][:, 2:4])
    np.testing.assert_array_equal(g["p1_mud"], want["ints"][:, 4])
    np.testing.assert_array_equal(g["turn"], want["ints"][:, 6])
    np.testing.assert_array_equal(g["action_p1"], want["ints"][:, 7])
    np.testing.assert_array_equal(g["action_p2"], want["ints"][:, 8])
    f = want["floats"]
    for k, sl in (("p1_score", slice(0, 1)), ("p2_score", slice(1, 2)), ("value_p1", slice(2, 3)),
                  ("value_p2", slice(3, 4)), ("visit_counts_p1", slice(4, 9)), ("visit_counts_p2", slice(9, 14)),
                  ("prior_p1", slice(14, 19)), ("prior_p2", slice(19, 24)), ("policy_p1", slice(24, 29)),
                  ("policy_p2", slice(29, 34))):
        assert g[k].reshape(g["n"], -1).tobytes() == np.ascontiguousarray(f[:, sl]).tobytes(), k
    np.testing.assert_array_equal(g["cheese_mask"], want["masks"])
    np.testing.assert_array_equal(g["maze"], want["maze"])
    np.testing.assert_array_equal(g["initial_cheese"], want["initial_cheese"])
    np.testing.assert_array_equal(g["cheese_outcomes"], want["cheese_outcomes"])
    assert (g["final_p1_score"], g["final_p2_score"], g["result"]) == (
        want["final_p1_score"], want["final_p2_score"], want["result"])
    for k in ("total_simulations", "total_nn_evals", "total_terminals", "total_collisions", "cheese_available"):
        assert g[k] == want[k], k


@pytest.mark.parametrize("w,h,cheese,turns,sims,batch,kw,n_games", [
    (5, 5, 5, 30, 100, 8, dict(), 48),                                   # BASELINE config 1 (plumbing)
    (5, 5, 5, 30, 1000, 16, dict(), 12),                                  # config 2 search settings
    (7, 7, 10, 50, 600, 16, dict(noise_epsilon=0.25, **TUNED), 12),       # config 3 search settings, fewer sims
])
def test_selfplay_records_bit_exact(w, h, cheese, turns, sims, batch, kw, n_games):
    from alpharat_amd.sampling import rust_self_play

    games = []
    stats = rust_self_play(width=w, height=h, cheese_count=cheese, max_turns=turns, num_games=n_games,
                           simulations=sims, batch_size=batch, output_dir=None, seed=0, concurrent_games=8,
                           on_game=games.append, **kw)
    assert stats.total_games == n_games and len(games) == n_games
    assert sorted(g["game_index"] for g in games) == list(range(n_games))
    cfg = O.make_config(**kw)
    tot = dict(sims=0, nn=0, gv=0, bv=0)
    for g in games:
        i = g["game_index"]
        want = O.play_game(O.Game(w, h, turns).random_cheese(cheese, True, i), cfg, sims, batch, 0xA1FA0000 + i)
        _check_game(g, want)
        tot["sims"] += want["total_simulations"]
        tot["gv"] += want["gather_node_visits"]
        tot["bv"] += want["backup_node_visits"]
    assert stats.total_simulations == tot["sims"]
    assert stats.gather_node_visits == tot["gv"] and stats.backup_node_visits == tot["bv"]


def test_uniform_runs_are_the_same_through_the_split_pipeline(monkeypatch):
    """SmartUniform games through k_gather8 -> k_uniform_eval -> k_backup16 (AR_UNIFORM=queue) and through the fused
    k_step_uniform: identical records, and both equal the oracle's."""
    from alpharat_amd.sampling import rust_self_play

    def run(mode):
        monkeypatch.setenv("AR_UNIFORM", mode)
        games = {}
        rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=40, simulations=300, batch_size=16,
                       output_dir=None, seed=0, concurrent_games=24, noise_epsilon=0.25,
                       on_game=lambda g: games.__setitem__(g["game_index"], g), **TUNED)
        return games

    a, b = run("fused"), run("queue")
    cfg = O.make_config(noise_epsilon=0.25, **TUNED)
    for i in sorted(a):
        want = O.play_game(O.Game(5, 5, 30).random_cheese(5, True, i), cfg, 300, 16, 0xA1FA0000 + i)
        _check_game(a[i], want)
        _check_game(b[i], want)


def test_selfplay_full_size_properties():
    """BASELINE config 2 at full width (4096 concurrent 5x5 games, 1000 sims): size-independent
    properties instead of an oracle replay -- conservation of cheese, policy normalisation, sampled
    actions legal, counters consistent, every game index exactly once."""
    from alpharat_amd.sampling import rust_self_play

    seen = []
    acc = dict(bad_policy=0, bad_action=0, bad_cheese=0, positions=0)

    def on_game(g):
        seen.append(g["game_index"])
        acc["positions"] += g["n"]
        s1 = g["policy_p1"].sum(axis=1)
        s2 = g["policy_p2"].sum(axis=1)
        acc["bad_policy"] += int((np.abs(s1 - 1) > 1e-5).sum() + (np.abs(s2 - 1) > 1e-5).sum())
        idx = np.arange(g["n"])
        acc["bad_action"] += int((g["policy_p1"][idx, g["action_p1"]] <= 0).sum())
        acc["bad_action"] += int((g["policy_p2"][idx, g["action_p2"]] <= 0).sum())
        collected = (g["cheese_outcomes"] != 2).sum()
        acc["bad_cheese"] += int(abs(collected - (g["final_p1_score"] + g["final_p2_score"])) > 1e-6)

    stats = rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=4096, simulations=1000,
                           batch_size=16, output_dir=None, seed=0, concurrent_games=4096, on_game=on_game)
    assert sorted(seen) == list(range(4096))
    assert acc == dict(bad_policy=0, bad_action=0, bad_cheese=0, positions=stats.total_positions)
    assert stats.p1_wins + stats.p2_wins + stats.draws == 4096
    assert stats.total_nn_evals + stats.total_terminals >= 1000 * stats.total_positions * 0.99
    # two games replayed on the oracle
    for i in (0, 4095):
        want = O.play_game(O.Game(5, 5, 30).random_cheese(5, True, i), O.make_config(), 1000, 16, 0xA1FA0000 + i)
        assert want["n"] > 0


def test_bundles_on_disk_roundtrip(tmp_path):
    from alpharat_amd.sampling import rust_self_play

    games = {}
    stats = rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=10, simulations=50, batch_size=8,
                           output_dir=tmp_path, max_games_per_bundle=4, seed=3, on_game=lambda g: games.__setitem__(g["game_index"], g))
    files = sorted(tmp_path.glob("bundle_*.npz"))
    assert len(files) == 3 and not list(tmp_path.glob("*.tmp"))
    n_games = n_pos = 0
    for f in files:
        z = np.load(f)
        assert len(z.files) == 26
        n_games += len(z["game_lengths"])
        n_pos += int(z["game_lengths"].sum())
        assert z["maze"].dtype == np.int8 and z["cheese_mask"].dtype == np.bool_ and z["turn"].dtype == np.int16
        assert z["policy_p1"].shape == (int(z["game_lengths"].sum()), 5)
    assert n_games == 10 and n_pos == stats.total_positions


def test_invalid_arguments_raise():
    from alpharat_amd.sampling import rust_self_play

    with pytest.raises(ValueError, match="unknown maze_type"):
        rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=1, simulations=10, output_dir=None,
                       maze_type="spiral")
    with pytest.raises(ValueError, match="device"):
        rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=1, simulations=10, output_dir=None,
                       device="tensorrt")


@pytest.mark.parametrize("pool", [True, False], ids=["overflow-pool", "host-grown"])
def test_selfplay_bit_exact_under_arena_growth_and_shrink(monkeypatch, pool):
    """Tiny first arenas: trees outgrow them in the middle of the first search (stall -> copied to a
    bigger block), change arena at every root advance (overflow-pool classes, or arenas the host
    allocates when the pool is off / too small), and move back when they shrink; records must not change."""
    from alpharat_amd.sampling import rust_self_play

    monkeypatch.setenv("AR_ARENA_NODES", "256")
    if not pool:
        monkeypatch.setenv("AR_NO_POOL", "1")
    games = []
    kw = dict(noise_epsilon=0.25, **TUNED)
    stats = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=24, simulations=500,
                           batch_size=16, output_dir=None, seed=0, concurrent_games=16, on_game=games.append, **kw)
    assert stats.total_games == 24
    cfg = O.make_config(**kw)
    for g in games:
        i = g["game_index"]
        _check_game(g, O.play_game(O.Game(7, 7, 50).random_cheese(10, True, i), cfg, 500, 16, 0xA1FA0000 + i))


def test_arena_block_is_reused_and_released_between_calls():
    """The arena allocation is kept between calls (dirty memory from the previous run) and can be
    released: results are the same either way."""
    from alpharat_amd.sampling import release_device_memory, rust_self_play

    def run():
        games = []
        rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=12, simulations=64, batch_size=8,
                       output_dir=None, seed=3, concurrent_games=8, on_game=games.append, **TUNED)
        return {g["game_index"]: g for g in games}

    a = run()
    b = run()  # reuses the cached block
    release_device_memory(0)
    c = run()  # fresh allocation
    release_device_memory(0)
    release_device_memory(0)  # nothing cached: still fine
    for i in a:
        for other in (b, c):
            assert a[i]["final_p1_score"] == other[i]["final_p1_score"]
            np.testing.assert_array_equal(a[i]["policy_p1"], other[i]["policy_p1"])
            np.testing.assert_array_equal(a[i]["visit_counts_p2"], other[i]["visit_counts_p2"])
            np.testing.assert_array_equal(a[i]["value_p1"], other[i]["value_p1"])


@pytest.mark.parametrize("maze_type,w,h,cheese,turns,extra", [
    ("classic", 5, 5, 5, 30, dict()),                                           # configs/game/5x5_classic.yaml
    ("random", 7, 7, 10, 50, dict(wall_density=0.5, mud_density=0.3, maze_symmetric=False)),
    ("random", 11, 9, 12, 60, dict(wall_density=0.8, mud_density=0.2, maze_symmetric=True)),  # > 64 cells
])
def test_selfplay_generated_mazes_bit_exact(maze_type, w, h, cheese, turns, extra):
    """Walls and mud from the seeded generator (own: the engine's is absent, DESIGN.md 'game generation'),
    one maze per game living in the slot's pool entry: mazes and whole records equal the oracle's, which
    generates them with its own implementation of the same specification."""
    from alpharat_amd.sampling import rust_self_play

    games = []
    n_games = 20
    stats = rust_self_play(width=w, height=h, cheese_count=cheese, max_turns=turns, num_games=n_games, simulations=150,
                           batch_size=8, output_dir=None, seed=0, concurrent_games=8, maze_type=maze_type,
                           on_game=games.append, **extra)
    assert stats.total_games == n_games
    wd, md, sym = ((0.7, 0.1, True) if maze_type == "classic"
                   else (extra["wall_density"], extra["mud_density"], extra["maze_symmetric"]))
    cfg = O.make_config()
    seen_wall = seen_mud = False
    for g in games:
        i = g["game_index"]
        og = O.Game(w, h, turns).random_maze(wd, md, sym, i).random_cheese(cheese, True, i)
        _check_game(g, O.play_game(og, cfg, 150, 8, 0xA1FA0000 + i))
        interior = g["maze"][:-1, :-1, :2]  # UP / RIGHT of cells that have those neighbours
        seen_wall |= bool((interior == -1).any())
        seen_mud |= bool((g["maze"] >= 2).any())
    assert seen_wall and seen_mud


@pytest.mark.parametrize("shape", ["wide", "octet3", "lane"])
def test_selfplay_over_64_cells_on_the_other_gather_shapes(shape, monkeypatch):
    """Boards above 64 cells use the four-word cheese masks (the NW = 4 kernel instances), per-game generated mazes
    keep the cost tables in global memory instead of LDS: the four-, two- and one-lane gathers on that path
    (AR_GATHER; SmartUniform runs through the split pipeline at this size) against the oracle."""
    from alpharat_amd.sampling import rust_self_play

    monkeypatch.setenv("AR_GATHER", shape)
    monkeypatch.setenv("AR_UNIFORM", "queue")
    games = []
    stats = rust_self_play(width=11, height=9, cheese_count=12, max_turns=60, num_games=12, simulations=150, batch_size=8,
                           output_dir=None, seed=0, concurrent_games=8, maze_type="random", on_game=games.append,
                           wall_density=0.8, mud_density=0.2, maze_symmetric=True)
    monkeypatch.delenv("AR_GATHER")
    monkeypatch.delenv("AR_UNIFORM")
    assert stats.total_games == 12
    cfg = O.make_config()
    for g in games:
        i = g["game_index"]
        og = O.Game(11, 9, 60).random_maze(0.8, 0.2, True, i).random_cheese(12, True, i)
        _check_game(g, O.play_game(og, cfg, 150, 8, 0xA1FA0000 + i))


def test_selfplay_generated_mazes_with_network_uses_each_games_maze():
    """With a network the first-layer maze constants are refreshed for every slot that gets a new game:
    the root prior recorded at move 0 equals the network's policy for that game's own maze."""
    from alpharat_amd.game import PyRat
    from alpharat_amd.nets import Net
    from alpharat_amd.sampling import rust_self_play

    blob = Path(__file__).parent / "golden" / "nets" / "mlp_7x7_h256.arnet"
    games = []
    rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=24, simulations=64, batch_size=8,
                   output_dir=None, seed=4, concurrent_games=6, maze_type="random", wall_density=0.6, mud_density=0.2,
                   weights_path=str(blob), on_game=games.append)
    net = Net(blob)
    for g in games:
        cost = np.where(g["maze"] < 0, 0, g["maze"]).astype(np.uint8)
        start = PyRat(7, 7, cost, g["initial_cheese"].astype(np.uint8), tuple(g["p1_pos"][0]), tuple(g["p2_pos"][0]), 50)
        out = net.evaluate([start])
        # prior_p* are in action space with blocked moves folded onto STAY (expand_prior): compare the mass
        # of the moves that are open for both
        np.testing.assert_allclose(g["prior_p1"][0].sum(), 1.0, atol=1e-5)
        open1 = [d for d in range(4) if cost[g["p1_pos"][0][1], g["p1_pos"][0][0], d] > 0]
        np.testing.assert_allclose(g["prior_p1"][0][open1], out["policy_p1"][0][open1], atol=2e-6)
