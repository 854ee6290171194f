#!/usr/bin/env python3
"""Benchmark of the self-play MCTS sampling path on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config 3 -- 7x7 open PyRat, 10 symmetric cheese, 50 turns,
`7x7_rust_tuned` search (1897 sims, c_puct .512, fpu .459, force_k .103, noise eps .25, batch 16),
PyRatMLP hidden 256 with seeded random weights (throughput does not depend on weight values),
synthetic seeded games. A "step" is one full pass of the hot path over one batch of `--games`
games per GPU: every game is played to the end (search -> sample -> move -> reuse tree, every turn).
Games are sharded over ranks with no collective in the data path (weak scaling: per-GPU games fixed).

value = MCTS simulations per second over all ranks (the reference's own count: the sum of root
visit totals, selfplay.rs:547); games/s, nn_evals/s and productive descents/s ride along as
extra keys. The timed region starts with everything resident on the device.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes of the tree kernels
B_NODE_VISIT, B_NEW_NODE, B_NN_LEAF = 300, 304, 1444
# HBM traffic of one k_gather launch at the default workload, from the PMC passes committed in
# profiles/r01_v6_pmc_hbm_traffic_default.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs):
# FETCH_SIZE 551 052 KB per dispatch, doubled as MI355X_MICROARCH.md prescribes for gfx950 (the factor is
# calibrated for wide streaming reads; these are scattered 16-byte reads, so it is an upper bound),
# WRITE_SIZE 540 052 KB per dispatch as read.
PMC_GATHER_FETCH_KB, PMC_GATHER_WRITE_KB = 551052.0, 540052.0
B_SELECT_VISIT = 204  # the select half of SURVEY 8d's 300 B node-visit (192 B read + 12 B virtual-loss writes)
B_LEAF_REQ = 40  # one evaluator request written by the gather: position + slot id

SEARCH = dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, noise_concentration=10.83)
GAME = dict(width=7, height=7, cheese_count=10, max_turns=50)
SIMS, BATCH = 1897, 16


def make_mlp_blob(path: Path) -> Path:
    """PyRatMLP(obs 349 -> 256 -> 256 -> heads) with seeded random weights, written as a weight blob."""
    import numpy as np

    from alpharat_amd.weights import write_blob

    if path.exists():
        return path
    rng = np.random.default_rng(0)
    d, h = 7 * 7 * 7 + 6, 256
    t = {}
    for name, (o, i) in {"trunk.0": (h, d), "trunk.4": (h, h)}.items():
        t[f"{name}.weight"] = (rng.standard_normal((o, i)) * np.sqrt(2.0 / i)).astype(np.float32)
        t[f"{name}.bias"] = np.zeros(o, np.float32)
    for bn in ("trunk.1", "trunk.5"):
        t[f"{bn}.weight"] = np.ones(h, np.float32)
        t[f"{bn}.bias"] = np.zeros(h, np.float32)
        t[f"{bn}.running_mean"] = np.zeros(h, np.float32)
        t[f"{bn}.running_var"] = np.ones(h, np.float32)
    for name, o in (("policy_p1_head", 5), ("policy_p2_head", 5), ("value_head", 2)):
        t[f"{name}.weight"] = (rng.standard_normal((o, h)) * 0.01).astype(np.float32)
        t[f"{name}.bias"] = np.zeros(o, np.float32)
    return write_blob(path, "mlp", 7, 7, t)


def cpu_baseline(blob: Path, evaluator: str) -> dict:
    """The oracle's self-play loop (one game per OS thread, the reference's worker structure) timed on
    this box's host cores over a bounded sample of the same workload."""
    import _oracle as O

    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    cfg = O.make_config(**SEARCH)
    net = O.Net(blob) if evaluator == "mlp" else None
    games = threads  # one game per thread: ~10-30 s of CPU work
    r = O.selfplay_bench(GAME["width"], GAME["height"], GAME["cheese_count"], GAME["max_turns"], games, cfg, SIMS, BATCH,
                         threads, backend=2 if net else 0, net=net)
    return {
        "value": r["simulations"] / r["elapsed_secs"], "unit": "simulations/s", "cores": threads, "kind": "port",
        "sample": f"{games} games of the same workload, {threads} threads, one game per thread "
                  f"({r['positions']} positions, {r['elapsed_secs']:.1f} s)",
        "games_per_sec": games / r["elapsed_secs"],
        "descents_per_sec": (r["nn_evals"] + r["terminals"]) / r["elapsed_secs"],
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=262144, help="games per GPU per step")
    ap.add_argument("--resident", type=int, default=65536,
                    help="games resident on the GPU at once (one lane each; finished games are replaced from the rest)")
    ap.add_argument("--evaluator", choices=["mlp", "uniform"], default="mlp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)

    if world > 1:
        # torch first: it ships its own copy of the HIP runtime, and a process can only bring up one --
        # loaded after torch, libalpharat_hip binds to that copy (tools/probe_torch_hip.py); the other
        # order leaves whichever runtime initialises second without a device
        import torch  # noqa: F401

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod

        # rehearsal knobs (one-GPU box): AR_BENCH_BACKEND=gloo AR_BENCH_DEVICE=0 run every rank on the same device
        backend = os.environ.get("AR_BENCH_BACKEND", "nccl")
        if "AR_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["AR_BENCH_DEVICE"])
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend)
        dist = dist_mod
        dist.barrier()
    from alpharat_amd.sampling import rust_self_play

    blob = make_mlp_blob(ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet") if rank == 0 else None
    if dist is not None:
        dist.barrier()
    blob = ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet"
    weights = str(blob) if args.evaluator == "mlp" else None

    def one_step(step_idx: int):
        # game ids are global and disjoint across ranks and steps; per-GPU work is fixed (weak scaling)
        first = (step_idx * world + rank) * args.games
        return rust_self_play(**GAME, num_games=args.games, simulations=SIMS, batch_size=BATCH, output_dir=None,
                              weights_path=weights, seed=0, first_game_index=first, concurrent_games=min(args.resident, args.games),
                              device_index=local_rank, **SEARCH)

    for w in range(args.warmup):
        one_step(-1 - w)

    from alpharat_amd import _lib

    def sync():
        # device-wide sync through the library (hipDeviceSynchronize) + torch's too when it is loaded
        _lib.check(_lib.load().ar_device_sync(local_rank))
        if dist is not None:
            import torch

            if dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    sync()
    t0 = time.perf_counter()
    stats = None
    dev_secs = gather_secs = 0.0
    dev_steps = gather_launches = 0
    for k in range(args.steps):
        s = one_step(k)
        stats = s if stats is None else stats + s
        # sequential passes: device times and launch counts add up (SelfPlayStats.__add__ merges parallel shards)
        dev_secs += s.device_secs
        dev_steps += s.steps
        gather_secs += s.gather_secs
        gather_launches += s.gather_launches
    sync()
    elapsed = time.perf_counter() - t0

    tot = dict(sims=stats.total_simulations, games=stats.total_games, nn=stats.total_nn_evals,
               desc=stats.total_nn_evals + stats.total_terminals, positions=stats.total_positions,
               nv=stats.gather_node_visits + stats.backup_node_visits, new=stats.new_nodes)
    tree_secs = dev_secs
    if dist is not None:
        import torch

        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor([float(x) for x in tot.values()], dtype=torch.float64, device=red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = dict(zip(tot.keys(), [float(x) for x in v.tolist()]))
    if rank != 0:
        return 0

    # roofline of the dominant kernel, k_gather (rank 0's own counters): algorithmic bytes of one launch =
    # SURVEY.md 8d's per-unit figures x the units one launch processes (node records inspected on the way
    # down, nodes created, leaf positions handed to the evaluator), divided by the launch's duration from
    # HIP events recorded on the kernel's own stream around every launch.
    gather_bytes = (B_SELECT_VISIT * stats.gather_node_visits + B_NEW_NODE * stats.new_nodes
                    + (B_LEAF_REQ * stats.total_nn_evals if args.evaluator == "mlp" else 0))
    launches = max(gather_launches, 1)
    if args.evaluator == "mlp" and gather_secs > 0:
        avg_launch_s = gather_secs / launches
        achieved = gather_bytes / launches / avg_launch_s / 1e9
        kernel = "k_gather"
    else:  # SmartUniform: one fused step kernel, timed as a whole
        gather_bytes += (B_NODE_VISIT - B_SELECT_VISIT) * stats.backup_node_visits
        launches = max(dev_steps, 1)
        avg_launch_s = tree_secs / launches
        achieved = gather_bytes / launches / max(avg_launch_s, 1e-12) / 1e9
        kernel = "k_step_uniform (+ k_advance)"
    step_bytes = (B_NODE_VISIT * (stats.gather_node_visits + stats.backup_node_visits) + B_NEW_NODE * stats.new_nodes
                  + (B_NN_LEAF * stats.total_nn_evals if args.evaluator == "mlp" else 0))
    out = {
        "metric": "MCTS simulations/sec, self-play 7x7 PyRat at the tuned 1897-sim config",
        "value": tot["sims"] / elapsed,
        "unit": "simulations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "7x7 open PyRat, 10 cheese, 50 turns, 7x7_rust_tuned (1897 sims, batch 16, noise 0.25), "
                               + ("PyRatMLP h256 random weights" if args.evaluator == "mlp" else "SmartUniform priors"),
                   "games_per_gpu_per_step": args.games, "resident_games_per_gpu": min(args.resident, args.games),
                   "parallelism": f"games sharded over {world} GPU(s), no collective"},
        "games_per_sec": tot["games"] / elapsed,
        "nn_evals_per_sec": tot["nn"] / elapsed,
        "descents_per_sec": tot["desc"] / elapsed,
        "avg_turns": tot["positions"] / max(tot["games"], 1),
        "node_visits_per_sec": tot["nv"] / elapsed,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # PMC counters cannot be read inside this process: the figure is the committed measurement of this
            # exact workload (same games / resident / evaluator), null for any other
            "traffic": ((2.0 * PMC_GATHER_FETCH_KB + PMC_GATHER_WRITE_KB) * 1024.0
                        if (kernel == "k_gather" and args.games == 262144 and min(args.resident, args.games) == 65536)
                        else None),
            "traffic_source": "profiles/r01_v6_pmc_hbm_traffic_default.txt (bytes per k_gather launch; FETCH_SIZE x2 + WRITE_SIZE)",
            "kernel": kernel, "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
            "algorithmic_bytes_per_launch": gather_bytes / launches,
            # the whole step (gather + evaluator + backup, tree reuse overlapped) for reference
            "step": {"batch_steps": dev_steps, "device_secs": tree_secs, "avg_step_ms": tree_secs / max(dev_steps, 1) * 1e3,
                     "algorithmic_bytes": step_bytes, "achieved_GBps": step_bytes / max(tree_secs, 1e-9) / 1e9},
        },
    }
    if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed on rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(blob, args.evaluator)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
