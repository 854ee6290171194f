"""-m gpu: the Searcher-level adaptor (alpharat_amd/searcher.py) -- `search(game)` as alpharat/ai/searcher_agent.py
calls it, `search_batch(games)` for tournament rounds -- against the oracle. With a network the oracle's search is
driven by the HIP evaluator through the callback backend (tests/test_gpu_pipeline_parity.py), so results are bit-exact."""
from pathlib import Path

import numpy as np
import pytest

import _oracle as O
from test_gpu_parity import TUNED, _games, _pyrat
from test_gpu_pipeline_parity import HipEvaluator

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden" / "nets"


def _same(got, want, name):
    for k in ("policy_p1", "policy_p2"):
        p = want[k].astype(np.float64)
        p = p / p.sum() if p.sum() > 0 else p  # searcher.py:97-105 renormalisation in f64
        assert getattr(got, k).dtype == np.float64 and np.array_equal(getattr(got, k), p), (name, k)
    for k in ("visit_counts_p1", "visit_counts_p2", "prior_p1", "prior_p2"):
        assert np.array_equal(getattr(got, k), want[k].astype(np.float64)), (name, k)
    assert (got.value_p1, got.value_p2, got.total_visits) == (float(want["value_p1"]), float(want["value_p2"]),
                                                            want["total_visits"]), name


def test_searcher_protocol_and_batch_uniform():
    from alpharat_amd.searcher import HipSearcher

    s = HipSearcher(simulations=300, batch_size=8, seed=21, **TUNED)
    items = list(_games())[:5]
    games = [_pyrat(og, mt) for _, og, mt in items]
    single = [s.search(g) for g in games]
    batch = s.search_batch(games)
    other_seeds = s.search_batch(games, seeds=[100 + i for i in range(len(games))])
    for i, (name, og, mt) in enumerate(items):
        want = O.search_once(og, O.make_config(**TUNED), 300, 8, seed=21)
        _same(single[i], want, name)
        _same(batch[i], want, name)
        _same(other_seeds[i], O.search_once(og, O.make_config(**TUNED), 300, 8, seed=100 + i), name)
    assert s.search_batch([]) == []


def test_searcher_batch_with_device_net_bit_exact_vs_oracle():
    """Tournament-style: several 7x7 positions searched at once with the device MLP; each equals the oracle's search
    of that position when its evaluate_batch is the same HIP evaluator."""
    from alpharat_amd.searcher import HipSearcher

    blob = GOLD / "mlp_7x7_h256.arnet"
    kw = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103)  # RustMCTSConfig.for_evaluation(): no noise
    s = HipSearcher(simulations=400, batch_size=16, seed=5, checkpoint=blob, **kw)
    ogs = []
    for i in range(6):
        og = O.Game(7, 7, 50).random_cheese(10, True, 40 + i)
        for d1, d2 in [(0, 2), (1, 3), (4, 4)][: i % 4]:
            og.make_move(d1, d2)
        ogs.append(og)
    got = s.search_batch([_pyrat(og, 50) for og in ogs])
    ev = HipEvaluator(blob, 7, 7, 50)
    for i, og in enumerate(ogs):
        want = O.search_once(og, O.make_config(**kw), 400, 16, seed=5, backend=4, net=ev.backend)
        _same(got[i], want, i)
    one = s.search(_pyrat(ogs[3], 50))
    _same(one, O.search_once(ogs[3], O.make_config(**kw), 400, 16, seed=5, backend=4, net=ev.backend), "single")


def test_predict_fn_from_checkpoint_matches_device_net():
    """make_batched_predict_fn (drop-in for alpharat/ai/predict_batch.py:21-77): the callback route and the
    device-resident route give the same search."""
    from alpharat_amd.searcher import HipSearcher, make_batched_predict_fn

    blob = GOLD / "mlp_5x5_h32.arnet"
    g = _pyrat(O.Game(5, 5, 30, cheese=[(2, 2), (1, 3), (3, 1), (0, 4), (4, 0)]), 30)
    fn = make_batched_predict_fn(blob)
    p1, p2, v1, v2 = fn([g, g])
    assert p1.shape == (2, 5) and p1.dtype == np.float32 and v1.shape == (2,) and abs(p1[0].sum() - 1) < 1e-5
    a = HipSearcher(simulations=96, batch_size=8, seed=3, predict_fn=fn).search(g)
    b = HipSearcher(simulations=96, batch_size=8, seed=3, checkpoint=blob).search(g)
    for k in ("policy_p1", "policy_p2", "visit_counts_p1", "prior_p1"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert (a.value_p1, a.total_visits) == (b.value_p1, b.total_visits)
