"""``rust_mcts_search`` on the MI355X: same signature, results and error behaviour as the PyO3
function (crates/alpharat-mcts/src/bindings.rs:228-304), served by ``ar_search`` in
libalpharat_hip.so. Installed under the reference's import name by ``alpharat_amd/shims/alpharat_mcts``.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Callable

import numpy as np

from . import _lib
from .game import Coordinates, PyRat, arrays_from_game


def spec_from_game(game: Any, keep: list) -> _lib.ArGameSpec:
    """ArGameSpec from any object with the PyRat attribute surface (game.pyi:240-300)."""
    cost, cheese = arrays_from_game(game)
    cost = np.ascontiguousarray(cost, dtype=np.uint8)
    cheese = np.ascontiguousarray(cheese, dtype=np.uint8)
    keep.extend([cost, cheese])
    p1, p2 = game.player1_position, game.player2_position
    return _lib.ArGameSpec(
        int(game.width), int(game.height), int(game.max_turns), int(game.turn),
        int(p1.x), int(p1.y), int(p2.x), int(p2.y),
        int(game.player1_mud_turns), int(game.player2_mud_turns),
        float(game.player1_score), float(game.player2_score),
        cost.ctypes.data_as(C.c_void_p), cheese.ctypes.data_as(C.c_void_p),
    )


def make_search_config(c_puct=1.5, fpu_reduction=0.2, force_k=2.0, noise_epsilon=0.0, noise_concentration=10.83,
                       collision_limit_min=1, collision_limit_max=256, collision_scaling_start=800,
                       collision_scaling_end=50_000, collision_scaling_power=1.0) -> _lib.ArSearchConfig:
    return _lib.ArSearchConfig(c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration, collision_limit_min,
                               collision_limit_max, collision_scaling_start, collision_scaling_end,
                               collision_scaling_power)


class SearchResult:
    """Attribute-compatible with PySearchResult (bindings.rs:26-99): float32[5] getters, floats, ints."""

    def __init__(self, r: _lib.ArSearchResult) -> None:
        f = lambda a: np.array(list(a), dtype=np.float32)  # noqa: E731
        self.policy_p1, self.policy_p2 = f(r.policy_p1), f(r.policy_p2)
        self.visit_counts_p1, self.visit_counts_p2 = f(r.visit_counts_p1), f(r.visit_counts_p2)
        self.prior_p1, self.prior_p2 = f(r.prior_p1), f(r.prior_p2)
        self.value_p1, self.value_p2 = float(r.value_p1), float(r.value_p2)
        self.total_visits, self.nn_evals = int(r.total_visits), int(r.nn_evals)
        self.terminals, self.collisions = int(r.terminals), int(r.collisions)

    def __repr__(self) -> str:
        return (f"SearchResult(total_visits={self.total_visits}, value_p1={self.value_p1:.4f}, "
                f"value_p2={self.value_p2:.4f})")


def _leaf_to_game(template: Any, cost: np.ndarray, leaf: _lib.ArLeaf) -> PyRat:
    w, h = int(template.width), int(template.height)
    cheese = np.zeros(w * h, dtype=np.uint8)
    for i in range(w * h):
        if (leaf.cheese_bits[i >> 6] >> (i & 63)) & 1:
            cheese[i] = 1
    total = int(round(float(template.player1_score) + float(template.player2_score) + len(template.cheese_positions())))
    return PyRat(w, h, cost, cheese, Coordinates(leaf.p1_x, leaf.p1_y), Coordinates(leaf.p2_x, leaf.p2_y),
                 int(template.max_turns), int(leaf.turn), float(leaf.p1_score), float(leaf.p2_score), int(leaf.p1_mud),
                 int(leaf.p2_mud), total)


def rust_mcts_search(game: Any, *, predict_fn: Callable | None = None, simulations: int = 100, batch_size: int = 8,
                     c_puct: float = 1.5, fpu_reduction: float = 0.2, force_k: float = 2.0, noise_epsilon: float = 0.0,
                     noise_concentration: float = 10.83, collision_limit_min: int = 1, collision_limit_max: int = 256,
                     collision_scaling_start: int = 800, collision_scaling_end: int = 50_000,
                     collision_scaling_power: float = 1.0, seed: int | None = None, net: Any = None,
                     device: int = 0) -> SearchResult:
    """Run MCTS on `game`. ``predict_fn(list[PyRat]) -> (policy_p1[N,5], policy_p2[N,5], value_p1[N],
    value_p2[N])`` with N <= batch_size, or None for smart-uniform priors. ``net`` (an
    ``alpharat_amd.nets.Net``) evaluates leaves on the device instead -- an extension."""
    L = _lib.load()
    keep: list = []
    spec = spec_from_game(game, keep)
    cfg = make_search_config(c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration, collision_limit_min,
                             collision_limit_max, collision_scaling_start, collision_scaling_end, collision_scaling_power)
    out = _lib.ArSearchResult()
    seed_p = C.pointer(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF)) if seed is not None else None
    err: list[BaseException] = []
    cb = _lib.ArPredictFn()
    if predict_fn is not None:
        cost = keep[0]

        def _cb(_user, leaves, n, pp1, pp2, pv1, pv2) -> int:
            try:
                games = [_leaf_to_game(game, cost, leaves[i]) for i in range(n)]
                r = predict_fn(games)
                p1 = np.ascontiguousarray(r[0], dtype=np.float32).reshape(n, 5)
                p2 = np.ascontiguousarray(r[1], dtype=np.float32).reshape(n, 5)
                v1 = np.ascontiguousarray(r[2], dtype=np.float32).reshape(n)
                v2 = np.ascontiguousarray(r[3], dtype=np.float32).reshape(n)
                C.memmove(pp1, p1.ctypes.data, n * 20)
                C.memmove(pp2, p2.ctypes.data, n * 20)
                C.memmove(pv1, v1.ctypes.data, n * 4)
                C.memmove(pv2, v2.ctypes.data, n * 4)
                return 0
            except BaseException as e:  # surfaced as RuntimeError below, like BackendError in the reference
                err.append(e)
                return 1

        cb = _lib.ArPredictFn(_cb)
    net_h = getattr(net, "handle", None) if net is not None else None
    rc = L.ar_search(C.byref(spec), C.byref(cfg), simulations, batch_size, seed_p, cb, None, net_h, device,
                     C.byref(out))
    if rc != 0 and err:
        raise RuntimeError(f"predict_fn raised an exception: {err[0]}") from err[0]
    _lib.check(rc)
    return SearchResult(out)


def search_many(games: list, *, simulations: int = 100, batch_size: int = 8, seeds: list[int] | None = None,
                net: Any = None, device: int = 0, **search_kwargs) -> list[SearchResult]:
    """Independent searches over many positions in one device run (evaluation-time callers)."""
    L = _lib.load()
    n = len(games)
    keep: list = []
    specs = (_lib.ArGameSpec * n)(*[spec_from_game(g, keep) for g in games])
    cfg = make_search_config(**search_kwargs)
    out = (_lib.ArSearchResult * n)()
    seed_arr = (C.c_uint64 * n)(*[s & 0xFFFFFFFFFFFFFFFF for s in seeds]) if seeds is not None else None
    net_h = getattr(net, "handle", None) if net is not None else None
    _lib.check(L.ar_search_many(specs, n, C.byref(cfg), simulations, batch_size, seed_arr, net_h, device, out))
    return [SearchResult(o) for o in out]
