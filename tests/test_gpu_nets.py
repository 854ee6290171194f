"""-m gpu: HIP encoder + policy/value heads through the C-ABI against the golden vectors made from
the reference's Python models and against the oracle. Tolerance 1e-5 (north star) on logits,
policies and values; the encoder is compared at the reference's own 1e-6 (tests/parity.rs:16)."""
import json
from pathlib import Path

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _game_from_obs(obs, w, h, max_turns=50):
    """Rebuild a position from a synthetic golden observation (tools/gen_net_golden.py synth_obs)."""
    from alpharat_amd.game import PyRat

    hw = w * h
    maze = obs[: hw * 4].reshape(h, w, 4)
    cost = np.where(maze < 0, 0, np.rint(maze * 10)).astype(np.uint8)
    p1 = int(np.argmax(obs[hw * 4: hw * 5]))
    p2 = int(np.argmax(obs[hw * 5: hw * 6]))
    cheese = (obs[hw * 6: hw * 7] > 0.5).astype(np.uint8)
    s = obs[hw * 7:]
    return PyRat(w, h, cost, cheese, (p1 % w, p1 // w), (p2 % w, p2 // w), max_turns, int(round(float(s[1]) * max_turns)),
                 float(s[4]) * 10, float(s[5]) * 10, int(round(float(s[2]) * 10)), int(round(float(s[3]) * 10)))


@pytest.mark.parametrize("path", sorted((GOLD / "encoder").glob("*.json")), ids=lambda p: p.stem)
def test_device_encoder_golden(path):
    from alpharat_amd.game import PyRat
    from alpharat_amd.nets import encode

    fx = json.loads(path.read_text())
    xy = lambda d: (d["x"], d["y"])
    g = PyRat.create_custom(fx["width"], fx["height"], walls=[(xy(w["pos1"]), xy(w["pos2"])) for w in fx["walls"]],
                            mud=[(xy(m["pos1"]), xy(m["pos2"]), m["value"]) for m in fx["mud"]],
                            cheese=[xy(c) for c in fx["cheese"]], player1_pos=xy(fx["p1_pos"]),
                            player2_pos=xy(fx["p2_pos"]), max_turns=fx["max_turns"])
    for d1, d2 in fx.get("moves", []):
        g.make_move(d1, d2)
    got = encode([g])[0]
    np.testing.assert_allclose(got, np.asarray(fx["expected"], np.float32), atol=1e-6, rtol=0)


@pytest.mark.parametrize("name", ["mlp_5x5_h32", "mlp_7x7_h256", "symmetric_5x5_h32", "symmetric_7x7_h256",
                                  "cnn_res_5x5_c16", "cnn_gpool_7x5_c16", "cnn_gpool_7x7_c64",
                                  "cnn_pooled_7x5_c16", "cnn_pooled_7x7_c32",  # (pooled: value_head.type "pooled")
                                  # boards above 8x8: matrix-core kernel with two row tiles per wavefront (c32, c64),
                                  # the FMA kernel's row / column loops (c16)
                                  "cnn_gpool_15x11_c32", "cnn_res_15x11_c64", "cnn_gpool_9x10_c16"])
def test_device_net_matches_reference_outputs(name):
    from alpharat_amd.nets import Net, encode

    gold = np.load(GOLD / "nets" / f"{name}.npz")
    w, h = (int(v) for v in next(t for t in name.split("_") if t[0].isdigit()).split("x"))
    games = [_game_from_obs(o, w, h) for o in gold["obs"]]
    np.testing.assert_allclose(encode(games), gold["obs"], atol=1e-6, rtol=0)
    out = Net(GOLD / "nets" / f"{name}.arnet").evaluate(games)
    for k in ("logits_p1", "logits_p2", "policy_p1", "policy_p2", "value_p1", "value_p2"):
        np.testing.assert_allclose(out[k], gold[k], atol=1e-5, rtol=1e-5, err_msg=f"{name}:{k}")


@pytest.mark.parametrize("name,w,h", [("symmetric_5x5_h32", 5, 5), ("symmetric_7x7_h256", 7, 7)])
def test_symmetric_on_matrix_cores_gives_the_bits_of_the_fma_loops(name, w, h, monkeypatch):
    """k_symmetric_mfma accumulates every output as the same k-ordered chain as k_symmetric's loops (AR_SYM_FMA=1),
    whatever tile row a leaf lands in: 70 positions (two full 32-leaf tiles and a ragged one), compared bit for bit,
    and again in another order."""
    from alpharat_amd.nets import Net

    gold = np.load(GOLD / "nets" / f"{name}.npz")
    games = [_game_from_obs(o, w, h) for o in gold["obs"]]
    games = (games * 3)[:70]
    net = Net(GOLD / "nets" / f"{name}.arnet")
    a = net.evaluate(games)
    b = net.evaluate(games[::-1])
    monkeypatch.setenv("AR_SYM_FMA", "1")
    c = net.evaluate(games)
    for k in a:
        assert a[k].tobytes() == c[k].tobytes(), k
        assert a[k].tobytes() == b[k][::-1].tobytes(), k


@pytest.mark.parametrize("name,w,h", [("cnn_gpool_7x7_c64", 7, 7), ("cnn_pooled_7x7_c32", 7, 7)])
def test_cnn_with_trunk_in_registers_gives_the_bits_of_the_three_image_kernel(name, w, h, monkeypatch):
    """k_cnn_mfma (trunk state in registers, one LDS image, pooling branch on the matrix cores) runs the same k-ordered
    chains as k_cnn (AR_CNN_LDS=1, chosen when the weights are loaded), whatever row a position lands in: 35 positions
    (ragged last tile), bit for bit, and again in another order."""
    from alpharat_amd.nets import Net

    gold = np.load(GOLD / "nets" / f"{name}.npz")
    games = [_game_from_obs(o, w, h) for o in gold["obs"]]
    games = (games * 2)[:35]
    net = Net(GOLD / "nets" / f"{name}.arnet")
    a = net.evaluate(games)
    b = net.evaluate(games[::-1])
    monkeypatch.setenv("AR_CNN_LDS", "1")
    c = Net(GOLD / "nets" / f"{name}.arnet").evaluate(games)
    for k in a:
        assert a[k].tobytes() == c[k].tobytes(), k
        assert a[k].tobytes() == b[k][::-1].tobytes(), k


def test_mlp_first_layer_variants_give_the_same_bits(monkeypatch):
    """k_mlp_mfma's first layer: the p1 / p2 weight rows summed into the start of the chain and only cheese and scalars on
    the matrix cores (default), the whole non-maze observation as a product with the operand staged in LDS (AR_MLP_FL=0)
    or formed in registers (AR_MLP_FL=1): the same k-ordered chain, the same bits; 150 positions."""
    from alpharat_amd.nets import Net

    gold = np.load(GOLD / "nets" / "mlp_7x7_h256.npz")
    games = [_game_from_obs(o, 7, 7) for o in gold["obs"]]
    games = (games * 7)[:150]
    net = Net(GOLD / "nets" / "mlp_7x7_h256.arnet")
    a = net.evaluate(games)
    for fl in ("0", "1"):
        monkeypatch.setenv("AR_MLP_FL", fl)
        c = net.evaluate(games)
        for k in a:
            assert a[k].tobytes() == c[k].tobytes(), (fl, k)


def test_search_with_device_net_close_to_oracle_net():
    """Search driven by the device MLP vs the oracle search driven by the oracle MLP. Network outputs
    agree to ~1e-6, not bit for bit, so this checks the plumbing (priors, values, visit totals), not
    bit-exactness."""
    from alpharat_amd.mcts import rust_mcts_search
    from alpharat_amd.nets import Net
    from alpharat_amd.game import PyRat

    blob = GOLD / "nets" / "mlp_5x5_h32.arnet"
    og = O.Game(5, 5, 30, cheese=[(2, 2), (1, 3), (3, 1), (0, 4), (4, 0)])
    want = O.search_once(og, O.make_config(), 64, 8, seed=11, backend=2, net=O.Net(blob))
    g = PyRat.create_custom(5, 5, cheese=[(2, 2), (1, 3), (3, 1), (0, 4), (4, 0)], max_turns=30)
    got = rust_mcts_search(g, simulations=64, batch_size=8, seed=11, net=Net(blob))
    assert got.total_visits == want["total_visits"] and got.nn_evals == want["nn_evals"]
    np.testing.assert_allclose(got.prior_p1, want["prior_p1"], atol=1e-5)
    np.testing.assert_allclose(got.prior_p2, want["prior_p2"], atol=1e-5)
    np.testing.assert_allclose(got.value_p1, want["value_p1"], atol=1e-3)


def test_selfplay_with_device_net_runs_and_is_consistent():
    from alpharat_amd.sampling import rust_self_play

    games = []
    stats = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=32, simulations=96, batch_size=16,
                           output_dir=None, seed=1, weights_path=str(GOLD / "nets" / "mlp_7x7_h256.arnet"),
                           c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, on_game=games.append)
    assert stats.total_games == 32 and stats.total_nn_evals > 0
    for g in games:
        assert np.all(np.abs(g["policy_p1"].sum(axis=1) - 1) < 1e-5)
        assert np.all(g["value_p1"] >= 0)  # softplus values, non-negative rewards


def test_selfplay_records_do_not_depend_on_scheduling(monkeypatch):
    """Groups of games pipelined on separate streams, tree reuse on a side stream, the gather round limit,
    the allocation steps per round and the lanes per wavefront only change when work happens, never what
    a game computes."""
    from alpharat_amd.sampling import rust_self_play

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        games = []
        rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=40, simulations=96, batch_size=16,
                       output_dir=None, seed=5, weights_path=str(GOLD / "nets" / "mlp_7x7_h256.arnet"), concurrent_games=24,
                       c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, on_game=games.append)
        for k in env:
            monkeypatch.delenv(k)
        return {g["game_index"]: g for g in games}

    base = run(AR_GROUPS=1)
    for env in (dict(AR_GROUPS=3), dict(AR_GROUPS=2, AR_GATHER_ROUNDS=5), dict(AR_ALLOC_PER_ROUND=1, AR_LANES_PER_WAVE=16),
                dict(AR_GROUPS=4, AR_ALLOC_PER_ROUND=7, AR_GATHER_ROUNDS=11), dict(AR_NO_ADVANCE_OVERLAP=1),
                dict(AR_GATHER="lane"), dict(AR_GATHER="octet"), dict(AR_GATHER="octet4", AR_GROUPS=2),
                dict(AR_GATHER="octet", AR_ALLOC_PER_ROUND=3, AR_NO_ADVANCE_OVERLAP=1),
                dict(AR_GATHER="octet3"), dict(AR_GATHER="octet3", AR_GROUPS=2),
                dict(AR_GATHER="wide"), dict(AR_GATHER="wide", AR_GROUPS=2), dict(AR_GATHER="wide", AR_GW_WAVES=1),
                dict(AR_GATHER="wide", AR_GW_WAVES=3, AR_GROUPS=3, AR_NO_ADVANCE_OVERLAP=1),
                dict(AR_GATHER="wide", AR_GW_PASSES=7), dict(AR_GATHER="wide", AR_GW_PASSES=23, AR_GROUPS=2, AR_GW_WAVES=2),
                dict(AR_GATHER="wide", AR_GW_PASSES=0, AR_STAGGER=1, AR_GROUPS=2), dict(AR_GW_PASSES=5, AR_TREE_GB=0.02),
                dict(AR_BACKUP="lane"), dict(AR_BACKUP="lane", AR_GATHER="lane"), dict(AR_BACKUP="group", AR_GATHER="lane", AR_GROUPS=2)):
        other = run(**env)
        assert sorted(other) == sorted(base)
        for i, g in base.items():
            for key, val in g.items():
                if isinstance(val, np.ndarray):
                    np.testing.assert_array_equal(val, other[i][key], err_msg=f"{env} game {i} {key}")
                else:
                    assert val == other[i][key], (env, i, key)


@pytest.mark.parametrize("hidden", [32, 48, 64, 96, 320], ids=lambda h: f"h{h}")
def test_device_mlp_other_hidden_widths_match_the_oracle_net(hidden, tmp_path):
    """Hidden widths the golden vectors do not cover take other kernel paths (32: matrix-core kernel whose activation rows
    are too short for the staged first-layer operand at 7x7 -- it is formed in registers; 48: scalar tile loops;
    64 / 96: matrix-core kernel with idle or single-tile wavefronts; 320: matrix-core second layer with
    two activation buffers). Seeded random weights, 70 random positions (more than one 64-leaf tile),
    compared with the oracle's forward pass of the same blob on the device's own observations."""
    from alpharat_amd.game import PyRat
    from alpharat_amd.nets import Net, encode
    from alpharat_amd.weights import write_blob

    rng = np.random.default_rng(hidden)
    w = h = 7
    d = w * h * 7 + 6
    t = {}
    for name, (o, i) in {"trunk.0": (hidden, d), "trunk.4": (hidden, hidden)}.items():
        t[f"{name}.weight"] = (rng.standard_normal((o, i)) * np.sqrt(2.0 / i)).astype(np.float32)
        t[f"{name}.bias"] = (rng.standard_normal(o) * 0.1).astype(np.float32)
    for bn in ("trunk.1", "trunk.5"):
        t[f"{bn}.weight"] = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32)
        t[f"{bn}.bias"] = (0.1 * rng.standard_normal(hidden)).astype(np.float32)
        t[f"{bn}.running_mean"] = (0.1 * rng.standard_normal(hidden)).astype(np.float32)
        t[f"{bn}.running_var"] = (1 + 0.1 * rng.random(hidden)).astype(np.float32)
    for name, o in (("policy_p1_head", 5), ("policy_p2_head", 5), ("value_head", 2)):
        t[f"{name}.weight"] = (rng.standard_normal((o, hidden)) * 0.2).astype(np.float32)
        t[f"{name}.bias"] = (0.1 * rng.standard_normal(o)).astype(np.float32)
    blob = write_blob(tmp_path / f"mlp_h{hidden}.arnet", "mlp", w, h, t)
    games = []
    for i in range(70):
        cells = rng.permutation(w * h)
        cheese = [(int(c % w), int(c // w)) for c in cells[: int(rng.integers(1, 14))]]
        g = PyRat.create_custom(w, h, cheese=cheese, player1_pos=(int(cells[20] % w), int(cells[20] // w)),
                                player2_pos=(int(cells[21] % w), int(cells[21] // w)), max_turns=50)
        for _ in range(int(rng.integers(0, 6))):
            g.make_move(int(rng.integers(0, 5)), int(rng.integers(0, 5)))
        games.append(g)
    obs = encode(games)
    want = O.Net(blob).forward(obs)
    got = Net(blob).evaluate(games)
    for k in ("logits_p1", "logits_p2", "policy_p1", "policy_p2", "value_p1", "value_p2"):
        np.testing.assert_allclose(got[k], want[k], atol=1e-5, rtol=1e-5, err_msg=f"h{hidden}:{k}")


def test_nn_eval_cache_skips_work_but_not_results():
    """cache_size > 0 (cached_backend.rs): repeated positions are served from the device table. A hit is
    bit for bit what the network computes again, so the records are those of the uncached run; every
    request is either a hit or a miss."""
    from alpharat_amd.sampling import rust_self_play

    def run(cache_size):
        games = []
        st = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=48, simulations=300,
                            batch_size=16, output_dir=None, seed=9, concurrent_games=32, cache_size=cache_size,
                            weights_path=str(GOLD / "nets" / "mlp_7x7_h256.arnet"), c_puct=0.512, fpu_reduction=0.459,
                            force_k=0.103, noise_epsilon=0.25, on_game=games.append)
        return st, {g["game_index"]: g for g in games}

    st0, base = run(0)
    st1, cached = run(4096)
    assert st0.cache_hits == 0 and st0.cache_misses == 0
    assert st1.cache_hits > 0
    assert st1.cache_hits + st1.cache_misses == st1.total_nn_evals == st0.total_nn_evals
    assert sorted(base) == sorted(cached)
    for i, g in base.items():
        for key, val in g.items():
            if isinstance(val, np.ndarray):
                np.testing.assert_array_equal(val, cached[i][key], err_msg=f"game {i} {key}")
            else:
                assert val == cached[i][key], (i, key)


def test_full_config_self_play_is_the_same_on_every_runtime_path(monkeypatch):
    """BASELINE config 3 as the bench runs it (7x7, 1897 simulations, batch 16, tuned constants, noise, MLP
    h256), 480 games on 160 lanes: trees outgrow their first arenas, slots are refilled, tree reuse runs on
    the side stream. The same games on the plainest runtime path (no overflow pool -> host-grown arenas,
    tree reuse in stream order) must give identical records."""
    from alpharat_amd.sampling import rust_self_play

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        games = []
        st = rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=480, simulations=1897,
                            batch_size=16, output_dir=None, seed=11, concurrent_games=160,
                            weights_path=str(GOLD / "nets" / "mlp_7x7_h256.arnet"), c_puct=0.512, fpu_reduction=0.459,
                            force_k=0.103, noise_epsilon=0.25, noise_concentration=10.83, on_game=games.append)
        for k in env:
            monkeypatch.delenv(k)
        return st, {g["game_index"]: g for g in games}

    st_a, a = run()
    st_b, b = run(AR_NO_POOL=1, AR_NO_ADVANCE_OVERLAP=1)
    assert sorted(a) == sorted(b) == list(range(480))
    assert st_a.total_simulations == st_b.total_simulations and st_a.total_nn_evals == st_b.total_nn_evals
    for i, g in a.items():
        for key, val in g.items():
            if isinstance(val, np.ndarray):
                np.testing.assert_array_equal(val, b[i][key], err_msg=f"game {i} {key}")
            else:
                assert val == b[i][key], (i, key)
    # conservation + normalisation on every position
    for g in a.values():
        assert np.all(np.abs(g["policy_p1"].sum(axis=1) - 1) < 1e-5) and np.all(np.abs(g["policy_p2"].sum(axis=1) - 1) < 1e-5)
        assert abs((g["cheese_outcomes"] != 2).sum() - (g["final_p1_score"] + g["final_p2_score"])) < 1e-6
