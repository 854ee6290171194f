"""N>1 path on CPU: two gloo ranks shard a game list, each 'plays' its block (the device call is
replaced by a stub that reports which game indices it was given -- there is no GPU here), and the
merged statistics must equal a single-rank run."""
import os
import sys
from pathlib import Path

import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def test_shard_blocks_are_disjoint_and_cover():
    from alpharat_amd.multigpu import shard_games

    for n in (0, 1, 7, 16, 1000, 4097):
        for w in (1, 2, 3, 8):
            blocks = [shard_games(n, w, r) for r in range(w)]
            covered = [i for f, c in blocks for i in range(f, f + c)]
            assert covered == list(range(n)), (n, w)
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def _fake_self_play(*, num_games, first_game_index=0, **kw):
    from alpharat_amd.sampling import SelfPlayStats

    idx = range(first_game_index, first_game_index + num_games)
    return SelfPlayStats(total_games=num_games, total_positions=sum(10 + i % 7 for i in idx),
                         total_simulations=sum(1000 + i for i in idx), elapsed_secs=1.0 + 0.1 * first_game_index,
                         p1_wins=sum(i % 3 == 0 for i in idx), p2_wins=sum(i % 3 == 1 for i in idx),
                         draws=sum(i % 3 == 2 for i in idx), total_cheese_collected=0.5 * num_games,
                         total_cheese_available=10 * num_games, min_turns=min((10 + i % 7 for i in idx), default=0),
                         max_turns=max((10 + i % 7 for i in idx), default=0), total_nn_evals=7 * num_games,
                         total_terminals=3 * num_games, total_collisions=num_games)


def _worker(rank, world, port, n_games, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from alpharat_amd.multigpu import self_play_sharded
    from test_multigpu_sharding import _fake_self_play

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = self_play_sharded(num_games=n_games, dist=dist, self_play=_fake_self_play, width=5, height=5)
    Path(out_dir, f"r{rank}.txt").write_text(
        f"{s.total_games} {s.total_positions} {s.total_simulations} {s.p1_wins} {s.p2_wins} {s.draws} "
        f"{s.min_turns} {s.max_turns} {s.elapsed_secs:.3f} {s.total_cheese_collected}")
    dist.destroy_process_group()


def test_two_rank_gloo_merge_equals_single_run(tmp_path):
    n = 37
    port = 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (tmp_path / "r0.txt").read_text(), (tmp_path / "r1.txt").read_text()
    assert r0 == r1
    one = _fake_self_play(num_games=n)
    got = r0.split()
    assert [int(x) for x in got[:8]] == [one.total_games, one.total_positions, one.total_simulations, one.p1_wins,
                                         one.p2_wins, one.draws, one.min_turns, one.max_turns]
    assert float(got[8]) == pytest.approx(1.0 + 0.1 * 19, abs=1e-3)  # max over ranks (rank 1 starts at game 19)
    assert float(got[9]) == pytest.approx(0.5 * n)
