"""The reference's ``Searcher`` protocol (``alpharat/mcts/searcher.py:21-25``) and its ``RustSearcher``
(``:28-117``) on the MI355X, plus a batched entry point for evaluation-time callers.

``HipSearcher.search(game)`` is what ``alpharat/ai/searcher_agent.py:40-56`` calls once per move;
``HipSearcher.search_batch(games)`` runs the searches of many concurrent games (a tournament round,
``alpharat/eval/tournament.py``) as ONE device run over ``ar_search_many`` -- independent trees, one lane each,
leaves of all of them batched into the same evaluator launches.

Evaluator, in order of preference: ``net`` (an ``alpharat_amd.nets.Net``) or ``checkpoint`` (a ``.pt`` / ``.arnet``
path: loaded onto the device, no torch in the loop), else ``predict_fn`` (host callback per leaf batch, exactly the
reference's contract; single searches only), else smart-uniform priors.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Sequence

import numpy as np

from .mcts import rust_mcts_search, search_many


@dataclass
class SearchResult:
    """Field-compatible with ``alpharat/mcts/result.py:16-43`` (float64 arrays, policies renormalised)."""

    policy_p1: np.ndarray
    policy_p2: np.ndarray
    value_p1: float
    value_p2: float
    visit_counts_p1: np.ndarray
    visit_counts_p2: np.ndarray
    prior_p1: np.ndarray
    prior_p2: np.ndarray
    total_visits: int


def _canonical(r: Any) -> SearchResult:
    # searcher.py:97-117: f32 -> f64, renormalise the policies so numpy.random.choice accepts them
    p1 = np.asarray(r.policy_p1, dtype=np.float64)
    p2 = np.asarray(r.policy_p2, dtype=np.float64)
    s1, s2 = p1.sum(), p2.sum()
    if s1 > 0:
        p1 = p1 / s1
    if s2 > 0:
        p2 = p2 / s2
    return SearchResult(p1, p2, float(r.value_p1), float(r.value_p2), np.asarray(r.visit_counts_p1, dtype=np.float64),
                        np.asarray(r.visit_counts_p2, dtype=np.float64), np.asarray(r.prior_p1, dtype=np.float64),
                        np.asarray(r.prior_p2, dtype=np.float64), int(r.total_visits))


class HipSearcher:
    """Constructor arguments of ``RustSearcher`` (searcher.py:43-59) + ``net`` / ``checkpoint`` / ``device_index``."""

    def __init__(self, simulations: int, c_puct: float = 1.5, force_k: float = 2.0, fpu_reduction: float = 0.2,
                 batch_size: int = 8, noise_epsilon: float = 0.0, noise_concentration: float = 10.83,
                 collision_limit_min: int = 1, collision_limit_max: int = 256, collision_scaling_start: int = 800,
                 collision_scaling_end: int = 50_000, collision_scaling_power: float = 1.0,
                 predict_fn: Callable[..., Any] | None = None, seed: int | None = None, net: Any = None,
                 checkpoint: str | Path | None = None, device_index: int = 0) -> None:
        self._simulations, self._batch_size, self._seed = int(simulations), int(batch_size), seed
        self._kw = dict(c_puct=c_puct, fpu_reduction=fpu_reduction, force_k=force_k, noise_epsilon=noise_epsilon,
                        noise_concentration=noise_concentration, collision_limit_min=collision_limit_min,
                        collision_limit_max=collision_limit_max, collision_scaling_start=collision_scaling_start,
                        collision_scaling_end=collision_scaling_end, collision_scaling_power=collision_scaling_power)
        self._predict_fn = predict_fn
        self._device = int(device_index)
        if net is None and checkpoint is not None:
            from .nets import Net

            cp = Path(checkpoint)
            net = Net(cp, self._device) if cp.suffix == ".arnet" else Net.from_checkpoint(cp, self._device)
        self._net = net

    @classmethod
    def from_config(cls, mcts_config: Any, checkpoint: str | Path | None = None, device_index: int = 0,
                    seed: int | None = None) -> "HipSearcher":
        """From a ``RustMCTSConfig`` (``alpharat/mcts/config.py:69-90``) or anything with the same attributes;
        the device-resident counterpart of ``RustMCTSConfig.build_searcher(checkpoint, device)`` (:92-118)."""
        names = ("simulations", "c_puct", "force_k", "fpu_reduction", "batch_size", "noise_epsilon",
                 "noise_concentration", "collision_limit_min", "collision_limit_max", "collision_scaling_start",
                 "collision_scaling_end", "collision_scaling_power")
        return cls(**{n: getattr(mcts_config, n) for n in names}, checkpoint=checkpoint, device_index=device_index,
                   seed=seed)

    def search(self, game: Any) -> SearchResult:
        net = self._net
        r = rust_mcts_search(game, predict_fn=None if net is not None else self._predict_fn,
                             simulations=self._simulations, batch_size=self._batch_size, seed=self._seed, net=net,
                             device=self._device, **self._kw)
        return _canonical(r)

    def search_batch(self, games: Sequence[Any], seeds: Sequence[int] | None = None) -> list[SearchResult]:
        """One search per game, all in one device run. ``seeds``: one per game; default: this searcher's seed for
        every game (what a loop over ``search`` would use), or entropy when it has none."""
        games = list(games)
        if not games:
            return []
        if self._net is None and self._predict_fn is not None:
            return [self.search(g) for g in games]  # a host callback serves one tree at a time
        if seeds is None and self._seed is not None:
            seeds = [self._seed] * len(games)
        rs = search_many(games, simulations=self._simulations, batch_size=self._batch_size,
                         seeds=list(seeds) if seeds is not None else None, net=self._net, device=self._device, **self._kw)
        return [_canonical(r) for r in rs]


def make_batched_predict_fn(checkpoint_path: str | Path, device: str | int = 0) -> Callable[[list[Any]], tuple]:
    """Drop-in for ``alpharat/ai/predict_batch.py:21-77``: ``predict_fn(list[PyRat]) -> (policy_p1[N,5], policy_p2[N,5],
    value_p1[N], value_p2[N])`` float32, evaluated by the HIP network kernels instead of torch."""
    from .nets import Net

    idx = device if isinstance(device, int) else (int(device.split(":")[1]) if ":" in device else 0)
    cp = Path(checkpoint_path)
    net = Net(cp, idx) if cp.suffix == ".arnet" else Net.from_checkpoint(cp, idx)

    def predict_fn(games: list[Any]) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        o = net.evaluate(games)
        return o["policy_p1"], o["policy_p2"], o["value_p1"], o["value_p2"]

    predict_fn.net = net  # keeps the device weights alive with the closure
    return predict_fn
