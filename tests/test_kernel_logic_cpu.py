"""The product's device-side search logic (alpharat_amd/csrc/dev_*.h compiled for the CPU by
tests/hostsim) against the oracle: bit-exact trees, search results and whole-game records.
This checks control flow, arena reuse/growth and f32 operation order without a GPU; the same
comparisons run against the real HIP kernels under -m gpu."""
import numpy as np
import pytest

import _hostsim as H
import _oracle as O

TUNED = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103)


def games():
    yield "open5_corner", O.Game(5, 5, 100, p1=(0, 0), p2=(4, 4), cheese=[(2, 2), (1, 3), (3, 1)]), 100
    yield "same_cell", O.Game(5, 5, 100, p1=(2, 2), p2=(2, 2), cheese=[(i, 0) for i in range(5)]), 100
    yield "short", O.Game(5, 5, 3, p1=(0, 0), p2=(2, 0), cheese=[(1, 0)]), 3
    g = O.Game(5, 5, 100, p1=(2, 2), p2=(4, 4), cheese=[(0, 0), (4, 0)], mud=[((2, 2), (2, 3), 3)],
               walls=[((0, 0), (0, 1)), ((3, 3), (4, 3))])
    g.make_move(0, 4)
    yield "mud_wall", g, 100
    yield "7x7", O.Game(7, 7, 50).random_cheese(10, True, 5), 50


@pytest.mark.parametrize("sims,batch", [(1, 1), (40, 1), (200, 8), (600, 16)])
def test_single_search_tree_bit_exact(sims, batch):
    for name, g, mt in games():
        for cfgkw in (dict(), TUNED):
            cfg = O.make_config(**cfgkw)
            want = O.search_once(g, cfg, sims, batch, seed=42)
            got = H.run(g, mt, cfg, sims, batch, 42, single=True)
            assert got["error"] == 0, name
            wd = want["tree"].dump()
            assert got["dump_count"] == len(wd), name
            np.testing.assert_array_equal(got["dump"], wd, err_msg=name)
            f = got["last"]
            for k, sl in (("value_p1", 2), ("value_p2", 3)):
                assert np.float32(want[k]).tobytes() == f[sl].tobytes(), (name, k)
            for k, a in (("visit_counts_p1", 4), ("visit_counts_p2", 9), ("prior_p1", 14), ("prior_p2", 19),
                         ("policy_p1", 24), ("policy_p2", 29)):
                assert want[k].tobytes() == f[a:a + 5].tobytes(), (name, k)
            assert list(got["last_counts"]) == [want[k] for k in ("total_visits", "nn_evals", "terminals", "collisions")]
            assert got["gather_node_visits"] == int(want["counters"][0])
            assert got["backup_node_visits"] == int(want["counters"][1])
            assert got["new_nodes"] == int(want["counters"][2])


def _same_game(want, got):
    assert got["error"] == 0 and got["status"] == 2
    assert got["n"] == want["n"]
    np.testing.assert_array_equal(got["ints"], want["ints"])
    assert got["floats"].tobytes() == want["floats"].tobytes()
    np.testing.assert_array_equal(got["masks"], want["masks"])
    for k in ("total_simulations", "total_nn_evals", "total_terminals", "total_collisions", "gather_node_visits",
              "backup_node_visits", "new_nodes"):
        assert got[k] == want[k], k
    assert (got["final_p1_score"], got["final_p2_score"]) == (want["final_p1_score"], want["final_p2_score"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_whole_game_records_bit_exact_5x5(seed):
    g = O.Game(5, 5, 30).random_cheese(5, True, seed)
    cfg = O.make_config()
    want = O.play_game(g, cfg, 300, 8, 0xA1FA0000 + seed)
    _same_game(want, H.run(g, 30, cfg, 300, 8, 0xA1FA0000 + seed))


def test_whole_game_tuned_noise_7x7_and_split_eval_path():
    g = O.Game(7, 7, 50).random_cheese(10, True, 3)
    cfg = O.make_config(noise_epsilon=0.25, **TUNED)
    want = O.play_game(g, cfg, 400, 16, 77)
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77))
    # leaves stored + evaluated outside the gather (network / predict_fn path), same answers
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77, eval_mode=1))
    # the fused multi-batch machine of the SmartUniform step kernel, same answers
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77, eval_mode=2))
    # gather cut off every 7 rounds and resumed from its parked lane state
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77, eval_mode=3))
    # the work-queue gather (dev_gatherw.h), run pass by pass like a wavefront runs it; with uniform priors nearly
    # every allocation step draws a tie break, so entries wait for the depth-first order all the time
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77, eval_mode=4))
    # ... and with random cuts of its queue (a pass takes fewer items than it could)
    _same_game(want, H.run(g, 50, cfg, 400, 16, 77, eval_mode=5))


@pytest.mark.parametrize("sims,batch,seed", [(1897, 16, 77), (300, 8, 5), (150, 3, 9)])
def test_work_queue_gather_with_network_like_priors(sims, batch, seed):
    """The work-queue gather in the regime it is built for: priors and values that differ from outcome to outcome
    (a fixed hash of the position stands in for the network, on both sides), so ties are rare and the levels of a
    pick really run side by side. Whole-game records against the oracle: queue taken 64 at a time, with random cuts,
    and with a single position record per game (every other parent's record goes through the scratch area)."""
    g = O.Game(7, 7, 50).random_cheese(10, True, seed)
    cfg = O.make_config(noise_epsilon=0.25, **TUNED)
    want = O.play_game(g, cfg, sims, batch, seed, backend=4, net=O.CallbackBackend(H.hashed_eval(7)))
    assert want["total_nn_evals"] > 0
    _same_game(want, H.run(g, 50, cfg, sims, batch, seed, eval_mode=8))  # the lane-per-game gather on the same evaluator
    for mode in (6, 7, 9):
        got = H.run(g, 50, cfg, sims, batch, seed, eval_mode=mode)
        _same_game(want, got)
        assert got["wide_gathers"] > 0


def test_arena_growth_keeps_results():
    g = O.Game(7, 7, 50).random_cheese(10, True, 9)
    cfg = O.make_config(**TUNED)
    want = O.play_game(g, cfg, 500, 16, 5)
    got = H.run(g, 50, cfg, 500, 16, 5, arena_nodes=64)
    assert got["grows"] >= 1
    _same_game(want, got)


def test_constant_value_backend_search():
    # backend.rs:114-129 ConstantValueBackend: non-zero leaf values through backup
    g = O.Game(5, 5, 100, p1=(1, 1), p2=(3, 3), cheese=[(2, 2), (0, 4)])
    cfg = O.make_config()
    want = O.search_once(g, cfg, 120, 4, seed=123, backend=1, v1=1.5, v2=0.5)
    got = H.run(g, 100, cfg, 120, 4, 123, single=True, eval_mode=1, v1=1.5, v2=0.5)
    np.testing.assert_array_equal(got["dump"], want["tree"].dump())
