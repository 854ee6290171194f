"""Game sharding over the GPUs of one node: games are independent, so each
rank plays a contiguous block of game indices on its own device and nothing is exchanged while
sampling. The only cross-rank step is summing the run statistics afterwards."""
from __future__ import annotations

import os
from typing import Any

from .sampling import SelfPlayStats

_SUM_FIELDS = ("total_games", "total_positions", "total_simulations", "p1_wins", "p2_wins", "draws",
               "total_cheese_collected", "total_cheese_available", "total_nn_evals", "total_terminals",
               "total_collisions", "cache_hits", "cache_misses", "gather_node_visits", "backup_node_visits",
               "new_nodes")


def shard_games(num_games: int, world: int, rank: int) -> tuple[int, int]:
    """(first_game_index, count) of rank's block; blocks are contiguous, disjoint and cover 0..num_games."""
    base, extra = divmod(num_games, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def reduce_stats(stats: SelfPlayStats, dist: Any = None, device: str | None = None) -> SelfPlayStats:
    """Sum the counters over ranks (one all_reduce), max of elapsed, min/max of turns."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return stats
    import torch

    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    sums = torch.tensor([float(getattr(stats, k)) for k in _SUM_FIELDS], dtype=torch.float64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    mx = torch.tensor([stats.elapsed_secs, stats.device_secs, float(stats.steps), float(stats.max_turns)],
                      dtype=torch.float64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    mn = torch.tensor([float(stats.min_turns) if stats.total_games > 0 else float("inf")], dtype=torch.float64,
                      device=dev)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    kw = {k: (float(v) if k == "total_cheese_collected" else int(round(v))) for k, v in zip(_SUM_FIELDS, sums.tolist())}
    kw.update(elapsed_secs=mx[0].item(), device_secs=mx[1].item(), steps=int(mx[2].item()), max_turns=int(mx[3].item()),
              min_turns=0 if mn[0].item() == float("inf") else int(mn[0].item()))
    return SelfPlayStats(**kw)


def self_play_sharded(*, num_games: int, dist: Any = None, self_play=None, **kwargs: Any) -> SelfPlayStats:
    """Run this rank's block and return the merged stats (identical on every rank)."""
    if self_play is None:
        from .sampling import rust_self_play as self_play
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    first, count = shard_games(num_games, world, rank)
    kwargs.setdefault("device_index", local)
    stats = self_play(num_games=count, first_game_index=kwargs.pop("first_game_index", 0) + first, **kwargs)
    return reduce_stats(stats, dist)
