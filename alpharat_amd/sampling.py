"""``rust_self_play`` / ``SelfPlayStats`` / ``SelfPlayProgress`` on the MI355X: the same keyword
arguments, attributes and error behaviour as the PyO3 module
(crates/alpharat-sampling/src/bindings.rs:28-201, 268-483), served by ``ar_selfplay_run``.
Installed under the reference's import name by ``alpharat_amd/shims/alpharat_sampling``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Any, Callable

import numpy as np

from . import _lib


class SelfPlayStats:
    """Read-only stats (bindings.rs:34-158): raw counters plus the derived rates."""

    _RAW = ("total_games", "total_positions", "total_simulations", "elapsed_secs", "p1_wins", "p2_wins", "draws",
            "total_cheese_collected", "total_cheese_available", "min_turns", "max_turns", "total_nn_evals",
            "total_terminals", "total_collisions", "cache_hits", "cache_misses",
            # extensions: roofline instrumentation
            "gather_node_visits", "backup_node_visits", "new_nodes", "device_secs", "steps", "gather_secs",
            "gather_launches")

    def __init__(self, s: _lib.ArSelfPlayStats | None = None, **kw: Any) -> None:
        for k in self._RAW:
            v = getattr(s, k) if s is not None else kw.get(k, 0)
            object.__setattr__(self, k, float(v) if k in ("elapsed_secs", "total_cheese_collected", "device_secs", "gather_secs") else int(v))

    def __setattr__(self, k: str, v: Any) -> None:
        raise AttributeError("SelfPlayStats is read-only")

    def _rate(self, num: float) -> float:
        return num / self.elapsed_secs if self.elapsed_secs > 0 else 0.0

    games_per_second = property(lambda s: s._rate(s.total_games))
    positions_per_second = property(lambda s: s._rate(s.total_positions))
    simulations_per_second = property(lambda s: s._rate(s.total_simulations))
    nn_evals_per_second = property(lambda s: s._rate(s.total_nn_evals))

    @property
    def cheese_utilization(self) -> float:
        return self.total_cheese_collected / self.total_cheese_available if self.total_cheese_available > 0 else 0.0

    @property
    def avg_turns(self) -> float:
        return self.total_positions / self.total_games if self.total_games > 0 else 0.0

    @property
    def draw_rate(self) -> float:
        return self.draws / self.total_games if self.total_games > 0 else 0.0

    @property
    def nn_eval_fraction(self) -> float:
        return self.total_nn_evals / self.total_simulations if self.total_simulations > 0 else 0.0

    @property
    def terminal_fraction(self) -> float:
        return self.total_terminals / self.total_simulations if self.total_simulations > 0 else 0.0

    @property
    def collision_fraction(self) -> float:
        t = self.total_nn_evals + self.total_terminals + self.total_collisions
        return self.total_collisions / t if t > 0 else 0.0

    @property
    def cache_hit_rate(self) -> float:
        t = self.cache_hits + self.cache_misses
        return self.cache_hits / t if t > 0 else 0.0

    def __add__(self, o: "SelfPlayStats") -> "SelfPlayStats":
        """Merge shard stats (multi-GPU): sums, min/max of turns, max of elapsed."""
        kw = {k: getattr(self, k) + getattr(o, k) for k in self._RAW}
        kw["elapsed_secs"] = max(self.elapsed_secs, o.elapsed_secs)
        kw["device_secs"] = max(self.device_secs, o.device_secs)
        kw["steps"] = max(self.steps, o.steps)
        kw["gather_secs"] = max(self.gather_secs, o.gather_secs)
        kw["gather_launches"] = max(self.gather_launches, o.gather_launches)
        mins = [s.min_turns for s in (self, o) if s.total_games > 0]
        kw["min_turns"] = min(mins) if mins else 0
        kw["max_turns"] = max(self.max_turns, o.max_turns)
        return SelfPlayStats(**kw)

    def __repr__(self) -> str:
        return (f"SelfPlayStats(games={self.total_games}, positions={self.total_positions}, "
                f"sims={self.total_simulations}, elapsed={self.elapsed_secs:.2f}s, "
                f"sims/s={self.simulations_per_second:.0f})")


class SelfPlayProgress:
    """Live counters readable from another thread while ``rust_self_play`` runs (bindings.rs:167-201)."""

    def __init__(self) -> None:
        self._c = _lib.ArProgress()

    games_completed = property(lambda s: int(s._c.games_completed))
    positions_completed = property(lambda s: int(s._c.positions_completed))
    simulations_completed = property(lambda s: int(s._c.simulations_completed))
    nn_evals_completed = property(lambda s: int(s._c.nn_evals_completed))


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def record_to_dict(v: _lib.ArGameRecordView) -> dict:
    """Copy one finished game out of the sink callback (the view dies when the callback returns)."""
    n, hw = int(v.n_positions), int(v.width) * int(v.height)
    d = dict(
        width=int(v.width), height=int(v.height), max_turns=int(v.max_turns), game_index=int(v.game_index), n=n,
        maze=_arr(v.maze, hw * 4, np.int8).reshape(v.height, v.width, 4),
        initial_cheese=_arr(v.initial_cheese, hw, np.uint8).reshape(v.height, v.width),
        cheese_outcomes=_arr(v.cheese_outcomes, hw, np.uint8).reshape(v.height, v.width),
        final_p1_score=float(v.final_p1_score), final_p2_score=float(v.final_p2_score), result=int(v.result),
        cheese_available=int(v.cheese_available), total_simulations=int(v.total_simulations),
        total_nn_evals=int(v.total_nn_evals), total_terminals=int(v.total_terminals),
        total_collisions=int(v.total_collisions),
        p1_pos=_arr(v.p1_pos, n * 2, np.uint8).reshape(n, 2), p2_pos=_arr(v.p2_pos, n * 2, np.uint8).reshape(n, 2),
        p1_score=_arr(v.p1_score, n, np.float32), p2_score=_arr(v.p2_score, n, np.float32),
        p1_mud=_arr(v.p1_mud, n, np.uint8), p2_mud=_arr(v.p2_mud, n, np.uint8), turn=_arr(v.turn, n, np.uint16),
        cheese_mask=_arr(v.cheese_mask, n * hw, np.uint8).reshape(n, hw),
        value_p1=_arr(v.value_p1, n, np.float32), value_p2=_arr(v.value_p2, n, np.float32),
        visit_counts_p1=_arr(v.visit_counts_p1, n * 5, np.float32).reshape(n, 5),
        visit_counts_p2=_arr(v.visit_counts_p2, n * 5, np.float32).reshape(n, 5),
        prior_p1=_arr(v.prior_p1, n * 5, np.float32).reshape(n, 5), prior_p2=_arr(v.prior_p2, n * 5, np.float32).reshape(n, 5),
        policy_p1=_arr(v.policy_p1, n * 5, np.float32).reshape(n, 5),
        policy_p2=_arr(v.policy_p2, n * 5, np.float32).reshape(n, 5),
        action_p1=_arr(v.action_p1, n, np.uint8), action_p2=_arr(v.action_p2, n, np.uint8),
    )
    return d


def _resolve_weights(weights_path, onnx_model_path):
    """``onnx_model_path`` is accepted for drop-in compatibility: a path ending in ``.onnx`` is mapped to the
    weight blob next to it (``.arnet``; written from the ``.pt`` checkpoint next to it when missing). The ONNX
    graph itself is never read or executed."""
    if weights_path is None and onnx_model_path is not None:
        p = Path(onnx_model_path)
        cand = p if p.suffix == ".arnet" else p.with_suffix(".arnet")
        if p.with_suffix(".pt").exists():
            # the trainer overwrites the checkpoint in place: checkpoint_to_blob rewrites the blob when the checkpoint
            # is newer and is a cheap cache hit otherwise (a bare blob is only trusted when no checkpoint sits next to it)
            from .weights import checkpoint_to_blob

            cand = checkpoint_to_blob(p.with_suffix(".pt"))
        if not cand.exists():
            raise RuntimeError(f"no weight blob for {onnx_model_path}: expected {cand} "
                               "(alpharat_amd.weights.checkpoint_to_blob writes it from the .pt checkpoint)")
        weights_path = str(cand)
    return weights_path


def _params(*, width, height, cheese_count, max_turns, num_games, cheese_symmetric, maze_type, positions, wall_density,
            mud_density, maze_symmetric, simulations, batch_size, c_puct, fpu_reduction, force_k, noise_epsilon,
            noise_concentration, collision_limit_min, collision_limit_max, collision_scaling_start,
            collision_scaling_end, collision_scaling_power, num_threads, output_dir, max_games_per_bundle,
            weights_path, device, mux_max_batch_size, cache_size, seed, rng_seed_base, first_game_index,
            concurrent_games, device_index):
    if device_index is None:
        device_index = int(os.environ.get("LOCAL_RANK", "0")) if device in ("auto", "hip") else 0
    if output_dir is not None:
        Path(output_dir).mkdir(parents=True, exist_ok=True)
    cfg = _lib.ArSearchConfig(c_puct, fpu_reduction, force_k, noise_epsilon, noise_concentration, collision_limit_min,
                              collision_limit_max, collision_scaling_start, collision_scaling_end,
                              collision_scaling_power)
    enc = lambda s: None if s is None else str(s).encode()  # noqa: E731
    has_seed = seed is not None
    return _lib.ArSelfPlayParams(
        width, height, cheese_count, max_turns, num_games, int(cheese_symmetric), enc(maze_type), enc(positions),
        wall_density, mud_density, int(maze_symmetric), simulations, batch_size, cfg, num_threads, enc(output_dir),
        max_games_per_bundle, enc(weights_path), enc(device), mux_max_batch_size, cache_size, int(has_seed),
        (seed or 0) & 0xFFFFFFFFFFFFFFFF,
        ((rng_seed_base if rng_seed_base is not None else (0xA1FA0000 + (seed or 0))) & 0xFFFFFFFFFFFFFFFF),
        first_game_index, concurrent_games, device_index,
    )


def rust_self_play(*, width: int, height: int, cheese_count: int, max_turns: int, num_games: int,
                   cheese_symmetric: bool = True, maze_type: str = "open", positions: str = "corners",
                   wall_density: float = 0.7, mud_density: float = 0.1, maze_symmetric: bool = True,
                   simulations: int, batch_size: int = 8, c_puct: float = 1.5, fpu_reduction: float = 0.2,
                   force_k: float = 2.0, noise_epsilon: float = 0.0, noise_concentration: float = 10.83,
                   collision_limit_min: int = 1, collision_limit_max: int = 256, collision_scaling_start: int = 800,
                   collision_scaling_end: int = 50_000, collision_scaling_power: float = 1.0, num_threads: int = 4,
                   output_dir: str | os.PathLike | None, max_games_per_bundle: int = 32,
                   onnx_model_path: str | None = None, device: str = "auto", mux_max_batch_size: int = 256,
                   cache_size: int = 0, progress: SelfPlayProgress | None = None,
                   # extensions (not in the reference signature)
                   weights_path: str | None = None, seed: int | None = None, rng_seed_base: int | None = None,
                   first_game_index: int = 0, concurrent_games: int = 0, device_index: int | None = None,
                   on_game: Callable[[dict], None] | None = None) -> SelfPlayStats:
    """Run self-play on one MI355X and write bundle ``.npz`` files to ``output_dir``.

    ``onnx_model_path`` is accepted for drop-in compatibility: a path ending in ``.onnx`` is mapped to
    the weight blob next to it (``.arnet``, written by ``alpharat_amd.weights.checkpoint_to_blob``);
    the ONNX graph itself is never executed."""
    L = _lib.load()
    weights_path = _resolve_weights(weights_path, onnx_model_path)
    p = _params(width=width, height=height, cheese_count=cheese_count, max_turns=max_turns, num_games=num_games,
                cheese_symmetric=cheese_symmetric, maze_type=maze_type, positions=positions, wall_density=wall_density,
                mud_density=mud_density, maze_symmetric=maze_symmetric, simulations=simulations, batch_size=batch_size,
                c_puct=c_puct, fpu_reduction=fpu_reduction, force_k=force_k, noise_epsilon=noise_epsilon,
                noise_concentration=noise_concentration, collision_limit_min=collision_limit_min,
                collision_limit_max=collision_limit_max, collision_scaling_start=collision_scaling_start,
                collision_scaling_end=collision_scaling_end, collision_scaling_power=collision_scaling_power,
                num_threads=num_threads, output_dir=output_dir, max_games_per_bundle=max_games_per_bundle,
                weights_path=weights_path, device=device, mux_max_batch_size=mux_max_batch_size, cache_size=cache_size,
                seed=seed, rng_seed_base=rng_seed_base, first_game_index=first_game_index,
                concurrent_games=concurrent_games, device_index=device_index)
    out = _lib.ArSelfPlayStats()
    sink = _lib.ArGameSink()
    if on_game is not None:
        sink = _lib.ArGameSink(lambda _u, v: on_game(record_to_dict(v.contents)))
    prog = C.byref(progress._c) if progress is not None else None
    _lib.check(L.ar_selfplay_run(C.byref(p), prog, sink, None, C.byref(out)))
    return SelfPlayStats(out)


UNBOUNDED = 0xFFFFFFFF


class SelfPlaySession:
    """The same run in slices (include/alpharat_hip.h: ar_selfplay_open / _step / _close): the device engine, its
    resident games and the supply of new games stay alive between ``step`` calls. ``num_games=UNBOUNDED`` never
    runs out of games. Takes rust_self_play's keyword arguments.

        with SelfPlaySession(width=7, ..., num_games=UNBOUNDED, concurrent_games=65536) as s:
            window = s.step(1024)     # SelfPlayStats of what happened inside these 1024 batch steps
    """

    def __init__(self, *, on_game: Callable[[dict], None] | None = None, progress: SelfPlayProgress | None = None,
                 onnx_model_path: str | None = None, **kw: Any) -> None:
        L = _lib.load()
        defaults = dict(cheese_symmetric=True, maze_type="open", positions="corners", wall_density=0.7, mud_density=0.1,
                        maze_symmetric=True, batch_size=8, c_puct=1.5, fpu_reduction=0.2, force_k=2.0, noise_epsilon=0.0,
                        noise_concentration=10.83, collision_limit_min=1, collision_limit_max=256,
                        collision_scaling_start=800, collision_scaling_end=50_000, collision_scaling_power=1.0,
                        num_threads=4, output_dir=None, max_games_per_bundle=32, weights_path=None, device="auto",
                        mux_max_batch_size=256, cache_size=0, seed=None, rng_seed_base=None, first_game_index=0,
                        concurrent_games=0, device_index=None)
        defaults.update(kw)
        defaults["weights_path"] = _resolve_weights(defaults["weights_path"], onnx_model_path)
        self._unbounded = int(defaults["num_games"]) >= UNBOUNDED
        p = _params(**defaults)
        self._sink = _lib.ArGameSink()
        if on_game is not None:
            self._sink = _lib.ArGameSink(lambda _u, v: on_game(record_to_dict(v.contents)))
        self._progress = progress
        prog = C.byref(progress._c) if progress is not None else None
        self._h = C.c_void_p()
        self.finished = False
        _lib.check(L.ar_selfplay_open(C.byref(p), prog, self._sink, None, C.byref(self._h)))

    def step(self, batch_steps: int) -> SelfPlayStats:
        """`batch_steps` simulate_batch steps for every resident game; returns the window's stats."""
        if not self._h:
            raise RuntimeError("session is closed")
        out = _lib.ArSelfPlayStats()
        fin = C.c_int(0)
        _lib.check(_lib.load().ar_selfplay_step(self._h, int(batch_steps), C.byref(out), C.byref(fin)))
        self.finished = bool(fin.value)
        return SelfPlayStats(out)

    def run_to_end(self) -> SelfPlayStats:
        if self._unbounded:
            raise ValueError("a session with an endless supply of games (num_games=UNBOUNDED) has no end: use step(n)")
        return self.step(UNBOUNDED)

    def info(self) -> dict:
        """What the library chose: resident games actually on the device (bounded by the memory the trees need), groups,
        gather kernel (0 lane, 1 octet, 2 work queue, 3 fused uniform), pass limit per gather launch, tree region bytes."""
        if not self._h:
            raise RuntimeError("session is closed")
        out = _lib.ArSessionInfo()
        _lib.check(_lib.load().ar_selfplay_info(self._h, C.byref(out)))
        return {k: (float(getattr(out, k)) if k == "tree_pages_per_game" else int(getattr(out, k))) for k, _ in out._fields_}

    def close(self) -> SelfPlayStats | None:
        if not self._h:
            return None
        out = _lib.ArSelfPlayStats()
        h, self._h = self._h, C.c_void_p()
        _lib.check(_lib.load().ar_selfplay_close(h, C.byref(out)))
        return SelfPlayStats(out)

    def __enter__(self) -> "SelfPlaySession":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def release_device_memory(device_index: int = 0) -> None:
    """Frees the tree-arena allocation the library keeps on the device between rust_self_play calls
    (include/alpharat_hip.h: ar_release_device_memory), e.g. before a training step that needs the HBM."""
    _lib.check(_lib.load().ar_release_device_memory(int(device_index)))


def preload_cuda_libs() -> None:
    """No-op: the reference calls this before non-CPU runs (rust_sampling.py:176-183)."""


def preload_tensorrt_libs() -> None:
    """No-op (see preload_cuda_libs)."""
