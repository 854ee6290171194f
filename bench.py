#!/usr/bin/env python3
"""Benchmark of the self-play MCTS sampling path on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config 3 -- 7x7 open PyRat, 10 symmetric cheese, 50 turns,
`7x7_rust_tuned` search (1897 sims, c_puct .512, fpu .459, force_k .103, noise eps .25, batch 16),
PyRatMLP hidden 256 with seeded random weights (throughput does not depend on weight values),
synthetic seeded games.

The sampler runs as ONE persistent session per GPU (ar_selfplay_open / _step / _close): `--resident`
games live on the device, every finished game is replaced at once from an endless seeded supply, exactly
like the reference's workers claim the next game from a queue (selfplay.rs:609-650, bench_selfplay.rs:213-270).
A "step" is one bounded slice of that run: `--batch-steps` passes of the hot path (gather -> evaluate ->
backup, + tree reuse for the games that moved) over all resident games. Warm-up steps bring the session to
its steady state (games at every stage of their life); the timed steps then measure it. Games are sharded
over ranks with no collective in the data path (weak scaling: per-GPU resident games fixed).

value = MCTS simulations per second over all ranks (the reference's own count: the sum of root visit
totals of the moves finished inside the timed region, selfplay.rs:547 -- visits kept by tree reuse count);
productive descents/s (evaluations + terminals, what the search actually walks) and NN evaluations/s ride
along, as do games/s. The timed region starts with everything resident on the device.

Extra, non-headline evaluators: --evaluator symmetric (BASELINE config 4: SymmetricMLP h256, 7x7_rust_strong,
2693 sims), --evaluator cnn (config 5: CNN + global pooling c64, 4096 sims), --evaluator uniform (tree kernels only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

T_START = time.perf_counter()
# one hardware queue per stream of the sampler's step pipeline; must be in the environment before anything (torch in
# multi-rank runs) initialises the HIP runtime
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes of the tree kernels
B_NODE_VISIT, B_NEW_NODE, B_NN_LEAF = 300, 304, 1444
B_SELECT_VISIT = 204  # the select half of SURVEY 8d's 300 B node-visit (192 B read + 12 B virtual-loss writes)
B_LEAF_REQ = 40  # one evaluator request written by the gather: position + slot id

GAME = dict(width=7, height=7, cheese_count=10, max_turns=50)
WORKLOADS = {
    # evaluator: (search kwargs, sims, batch, description)
    "mlp": (dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, noise_concentration=10.83), 1897, 16,
            "7x7_rust_tuned (1897 sims, batch 16, noise 0.25), PyRatMLP h256 random weights"),
    "uniform": (dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, noise_concentration=10.83), 1897, 16,
                "7x7_rust_tuned (1897 sims, batch 16, noise 0.25), SmartUniform priors"),
    "symmetric": (dict(c_puct=0.512, fpu_reduction=0.479, force_k=0.025, noise_epsilon=0.25, noise_concentration=10.83), 2693, 16,
                  "7x7_rust_strong (2693 sims, batch 16, noise 0.25), SymmetricMLP h256 random weights"),
    "cnn": (dict(c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25, noise_concentration=10.83), 4096, 16,
            "4096 sims (tuned constants, batch 16, noise 0.25), PyRatCNN c64 res,res,gpool(32) random weights"),
}
# sources whose change invalidates a committed PMC traffic measurement of the tree kernels
TRAFFIC_SOURCES = ("dev_search.h", "dev_gatherw.h", "dev_gather8.h", "dev_backup16.h", "dev_engine.h", "dev_rng.h", "slot_layout.h", "alpharat_hip.hip")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        h.update((ROOT / "alpharat_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def committed_traffic(kernel: str, evaluator: str, resident: int):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (tools/pmc_traffic.py
    writes profiles/traffic.json). Counters cannot be read from inside this process, so the figure is only
    reported when it was measured on exactly these kernel sources and this workload; otherwise null."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, None
    try:
        t = json.loads(f.read_text())
    except ValueError:
        return None, None
    if t.get("source_hash") != kernel_source_hash() or t.get("evaluator") != evaluator or t.get("resident") != resident:
        return None, f"profiles/traffic.json is stale (measured on sources {t.get('source_hash')}, {t.get('evaluator')}, " \
                     f"{t.get('resident')} resident)"
    k = t.get("kernels", {}).get(kernel)
    if not k:
        return None, None
    return k["bytes_per_launch"], t.get("source")


def make_mlp_blob(path: Path) -> Path:
    """PyRatMLP(obs 349 -> 256 -> 256 -> heads) with seeded random weights, written as a weight blob."""
    import numpy as np

    from alpharat_amd.weights import write_blob

    if path.exists():
        return path
    path.parent.mkdir(parents=True, exist_ok=True)
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


def weights_for(evaluator: str, rank: int) -> str | None:
    if evaluator == "uniform":
        return None
    if evaluator == "mlp":
        blob = ROOT / "gpurun_out" / "bench_mlp_7x7_h256.arnet"
        if rank == 0:
            make_mlp_blob(blob)
        return str(blob)
    # the seeded random networks of the parity tests (tools/gen_net_golden.py)
    name = {"symmetric": "symmetric_7x7_h256", "cnn": "cnn_gpool_7x7_c64"}[evaluator]
    return str(ROOT / "tests" / "golden" / "nets" / f"{name}.arnet")


def cpu_baseline(blob: str | None, evaluator: str, max_secs: float, threads: int | None = None) -> dict:
    """The oracle's self-play loop (the reference's worker structure: OS threads claiming games from an atomic
    counter, 16 games per thread like bench_selfplay.rs:213) timed on this box's host cores over a BOUNDED
    sample of the same workload: no thread claims a new game after `max_secs`. value = sum over threads of
    (simulations of the thread / time its last game ended), so the ragged end of the sample does not count."""
    import _oracle as O

    search, sims, batch, _ = WORKLOADS[evaluator]
    cores = os.cpu_count() or 1
    if threads is None:
        threads = cores  # every hardware thread of the box
    cfg = O.make_config(**search)
    net = O.Net(blob) if blob else None
    games = 16 * threads
    r = O.selfplay_bench(GAME["width"], GAME["height"], GAME["cheese_count"], GAME["max_turns"], games, cfg, sims, batch,
                         threads, backend=2 if net else 0, net=net, max_secs=max_secs)
    return {
        # simulations of the sample / its wall time (the ragged end counts against the CPU; the sum of per-thread rates,
        # which leaves it out, rides along)
        "value": r["simulations"] / r["elapsed_secs"], "thread_rate_sum": r["thread_rate_sum"], "unit": "simulations/s",
        "cores": threads, "kind": "port", "build": "g++ -O3 -march=native (oracle/Makefile)",
        "sample": f"{r['games']} games of the same workload finished by {threads} threads claiming from a queue of "
                  f"{games} until {max_secs:.0f} s had passed ({r['positions']} positions, {r['elapsed_secs']:.1f} s wall)",
        "wall_rate": r["simulations"] / r["elapsed_secs"],
        "games_per_sec": r["games"] / r["elapsed_secs"],
        "descents_per_sec": (r["nn_evals"] + r["terminals"]) / r["elapsed_secs"],
    }


def under_profiler() -> str | None:
    """rocprofv3 (and friends) preload a tool library into the process; legs that open a second session are skipped
    under it (round 2 lost a --pmc pass inside the HIP runtime in exactly such a leg)."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in pre or "roctracer" in pre or "rocprofiler" in pre:
        return "LD_PRELOAD=" + pre
    for k in os.environ:
        if k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_")):
            return k
    return None


def full_launch_leg(evaluator: str, resident: int, batch_steps: int, device_index: int) -> dict:
    """The extra leg: the same workload with the games as ONE group, i.e. one gather launch over all resident games per
    batch step. Runs in this process, after the timed session is closed and its device memory released (the group
    count is a per-session choice the library reads from AR_GROUPS when a session opens)."""
    from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession

    search, sims, batch, _ = WORKLOADS[evaluator]
    weights = weights_for(evaluator, 0)
    os.environ["AR_GROUPS"] = "1"
    try:
        return _full_launch_session(evaluator, resident, batch_steps, device_index, search, sims, batch, weights)
    finally:
        del os.environ["AR_GROUPS"]


def _full_launch_session(evaluator, resident, batch_steps, device_index, search, sims, batch, weights) -> dict:
    from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession

    with SelfPlaySession(**GAME, num_games=UNBOUNDED, simulations=sims, batch_size=batch, output_dir=None,
                         weights_path=weights, seed=0, first_game_index=1 << 27, concurrent_games=resident,
                         device_index=device_index, **search) as s1:
        for _ in range(4):
            s1.step(batch_steps)
        tf = time.perf_counter()
        w1 = s1.step(batch_steps)
        w1 = _sum_windows(w1, s1.step(batch_steps))
        dt1 = time.perf_counter() - tf
    b1 = (B_SELECT_VISIT * w1.gather_node_visits + B_NEW_NODE * w1.new_nodes + B_LEAF_REQ * w1.total_nn_evals)
    l1 = max(w1.gather_launches, 1)
    a1 = b1 / l1 / max(w1.gather_secs / l1, 1e-12) / 1e9
    return {"what": "same workload, AR_GROUPS=1: one gather launch over all resident games per batch step (2 timed steps "
                    "after 4 warm-up steps of a fresh session in this process, after the timed session was closed)",
            "achieved": a1, "frac": a1 / HBM_PEAK_GBS, "avg_launch_ms": w1.gather_secs / l1 * 1e3,
            "algorithmic_bytes_per_launch": b1 / l1, "simulations_per_sec": w1.total_simulations / dt1,
            "avg_step_ms": w1.device_secs / max(w1.steps, 1) * 1e3}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-steps", type=int, default=0,
                    help="passes of the hot path (gather -> evaluate -> backup) over all resident games per step "
                         "(default: 1024 for mlp / uniform, 128 for symmetric, 64 for cnn)")
    ap.add_argument("--resident", type=int, default=0,
                    help="games resident on each GPU (default: 131072; 16384 for cnn); the library may choose fewer "
                         "(device memory), the line reports what it used")
    ap.add_argument("--evaluator", choices=sorted(WORKLOADS), default="mlp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-secs", type=float, default=8.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-full-launch", action="store_true",
                    help="skip the extra leg that times the gather kernel launched over all resident games at once")
    ap.add_argument("--deadline", type=float, default=480.0,
                    help="seconds after process start at which the timed loop stops early and reports the steps done")
    ap.add_argument("--warmup-batch-steps", type=int, default=0,
                    help="warm up for this many batch steps instead of --warmup steps (the extra evaluators' lines: a game of "
                         "config 4 lasts about 2900 batch steps, of config 5 about 4400; a steady-state window starts after that)")
    ap.add_argument("--record", action="store_true",
                    help="also build every finished game's record on the host and write the bundles (the reference's "
                         "recording path) inside the timed region: a sink callback + output_dir on a tmpfs")
    ap.add_argument("--cpu-sweep", action="store_true",
                    help="CPU baseline at 1, 8, 32, 64 and all threads (bench_selfplay.rs:213-224) instead of one count")
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
            os.environ.setdefault("AR_MEM_FRACTION", f"{0.8 / world:.3f}")  # the ranks share one device's memory
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend)
        dist = dist_mod
        dist.barrier()
    from alpharat_amd import _lib
    from alpharat_amd.sampling import UNBOUNDED, SelfPlaySession

    search, sims, batch, workload = WORKLOADS[args.evaluator]
    if args.batch_steps <= 0:
        args.batch_steps = {"symmetric": 128, "cnn": 64}.get(args.evaluator, 1024)
    if args.resident <= 0:
        args.resident = 16384 if args.evaluator == "cnn" else 131072
    weights = weights_for(args.evaluator, rank)
    if dist is not None:
        dist.barrier()

    def sync():
        # device-wide sync through the library (hipDeviceSynchronize) + torch's too when it is loaded
        _lib.check(_lib.load().ar_device_sync(local_rank))
        if dist is not None:
            import torch

            if dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    # game ids are global and disjoint across ranks: rank r plays r * 2^28, r * 2^28 + 1, ... (weak scaling)
    rec_dir = None
    rec_games = [0]
    extra = {}
    if args.record:
        import tempfile

        rec_dir = tempfile.mkdtemp(prefix=f"ar_bench_rec_r{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        extra = dict(output_dir=rec_dir, on_game=lambda g: rec_games.__setitem__(0, rec_games[0] + 1))
    session = SelfPlaySession(**GAME, num_games=UNBOUNDED, simulations=sims, batch_size=batch,
                              weights_path=weights, seed=0, first_game_index=rank << 28, concurrent_games=args.resident,
                              device_index=local_rank, **{"output_dir": None, **extra}, **search)
    info = session.info()
    t_open = time.perf_counter() - T_START
    warm_done = 0
    if args.warmup_batch_steps > 0:
        while warm_done < args.warmup_batch_steps:
            n = min(1024, args.warmup_batch_steps - warm_done)
            session.step(n)
            warm_done += n
    else:
        for _ in range(args.warmup):
            session.step(args.batch_steps)
        warm_done = args.warmup * args.batch_steps

    sync()
    t0 = time.perf_counter()
    acc = None
    done = 0
    for k in range(args.steps):
        w = session.step(args.batch_steps)
        acc = w if acc is None else _sum_windows(acc, w)
        done += 1
        # insurance against a kill: a provisional line after every timed step (stderr + a file; stdout carries
        # exactly one line, the final one)
        _progress(rank, {"provisional": True, "steps_done": done, "value": acc.total_simulations * world / (time.perf_counter() - t0),
                         "unit": "simulations/s (this rank x world)"})
        if time.perf_counter() - T_START > args.deadline and done < args.steps:
            print(f"bench: deadline of {args.deadline:.0f} s reached after {done} of {args.steps} timed steps", file=sys.stderr)
            break
    sync()
    elapsed = time.perf_counter() - t0
    session.close()
    stats = acc
    if rec_dir is not None:
        import shutil

        rec_files = sum(len(fs) for _, _, fs in os.walk(rec_dir))
        shutil.rmtree(rec_dir, ignore_errors=True)

    tot = dict(sims=stats.total_simulations, games=stats.total_games, nn=stats.total_nn_evals,
               desc=stats.total_nn_evals + stats.total_terminals, positions=stats.total_positions,
               nv=stats.gather_node_visits + stats.backup_node_visits, new=stats.new_nodes, done=done)
    if dist is not None:
        import torch

        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor([float(x) for x in tot.values()], dtype=torch.float64, device=red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = dict(zip(tot.keys(), [float(x) for x in v.tolist()]))
        tot["done"] = tot["done"] / world
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    # roofline of the dominant kernel, k_gather (rank 0's own counters): algorithmic bytes of one launch =
    # SURVEY.md 8d's per-unit figures x the units one launch processes (node records inspected on the way
    # down, nodes created, leaf positions handed to the evaluator), divided by the launch's duration from
    # HIP events recorded on the kernel's own stream around every launch.
    has_net = args.evaluator != "uniform"
    gather_bytes = (B_SELECT_VISIT * stats.gather_node_visits + B_NEW_NODE * stats.new_nodes
                    + (B_LEAF_REQ * stats.total_nn_evals if has_net else 0))
    if has_net and stats.gather_secs > 0:
        launches = max(stats.gather_launches, 1)
        avg_launch_s = stats.gather_secs / launches
        # the library's default gather is the work-queue kernel; AR_GATHER=lane / octet* select the others
        sel = os.environ.get("AR_GATHER", "")
        kernel = "k_gather" if sel == "lane" else "k_gather8" if sel.startswith("octet") else "k_gatherw"
    else:  # SmartUniform: one fused step kernel, timed as a whole
        gather_bytes += (B_NODE_VISIT - B_SELECT_VISIT) * stats.backup_node_visits
        launches = max(stats.steps, 1)
        avg_launch_s = stats.device_secs / launches
        kernel = "k_step_uniform (+ k_advance)"
    achieved = gather_bytes / launches / max(avg_launch_s, 1e-12) / 1e9
    step_bytes = (B_NODE_VISIT * (stats.gather_node_visits + stats.backup_node_visits) + B_NEW_NODE * stats.new_nodes
                  + (B_NN_LEAF * stats.total_nn_evals if has_net else 0))
    traffic, traffic_source = committed_traffic(kernel if has_net else "k_step_uniform", args.evaluator, args.resident)
    out = {
        "metric": "MCTS simulations/sec, self-play 7x7 PyRat at the tuned 1897-sim config",
        "value": tot["sims"] / elapsed,
        "unit": "simulations/s",
        "n_gpus": world,
        "steps": int(tot["done"]),
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(tot["done"], 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "7x7 open PyRat, 10 cheese, 50 turns, " + workload,
                   "step": f"{args.batch_steps} passes of gather -> evaluate -> backup (+ tree reuse) over all resident games "
                           "of a persistent session; finished games are replaced at once from an endless seeded supply",
                   # what the library put on the device (it bounds the request by the memory the trees need), not the argument
                   "resident_games_per_gpu": info["resident_games"], "resident_games_requested": args.resident,
                   "groups": info["groups"], "gather_pass_limit": info["gather_pass_limit"],
                   "tree_region_GB": info["tree_region_bytes"] / 2 ** 30, "batch_steps_per_step": args.batch_steps,
                   "parallelism": f"games sharded over {world} GPU(s), no collective"},
        # the three rates side by side: `value` counts root visits per finished move, which includes the visits
        # a reused subtree brings along; descents are what the search actually walks; evaluations are network calls
        "simulations_per_sec": tot["sims"] / elapsed,
        "descents_per_sec": tot["desc"] / elapsed,
        "nn_evals_per_sec": tot["nn"] / elapsed,
        "games_per_sec": tot["games"] / elapsed,
        "positions_per_sec": tot["positions"] / elapsed,
        "avg_turns": tot["positions"] / max(tot["games"], 1),
        "node_visits_per_sec": tot["nv"] / elapsed,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # HBM bytes per launch from the PMC passes committed under profiles/ -- null unless they were taken on
            # exactly these kernel sources and this workload (counters cannot be read inside this process)
            "traffic": traffic, "traffic_source": traffic_source, "kernel_source_hash": kernel_source_hash(),
            "kernel": kernel, "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
            # from 8192 resident games up the games run as two groups pipelined against each other (one group's
            # evaluator beside the other's tree walks): a launch then covers half the games, overlapped with other
            # kernels. The same kernel launched over all 65536 games at once (AR_GROUPS=1): 2.65 ms, frac 0.050, at
            # 531 M simulations/s (profiles/r02_bench_1group.json) -- the pipelining buys throughput, not roofline.
            "launches_per_batch_step": launches / max(stats.steps, 1),
            "algorithmic_bytes_per_launch": gather_bytes / launches,
            # the whole step (gather + evaluator + backup, tree reuse overlapped) for reference
            "step": {"batch_steps": stats.steps, "device_secs": stats.device_secs,
                     "avg_step_ms": stats.device_secs / max(stats.steps, 1) * 1e3, "algorithmic_bytes": step_bytes,
                     "achieved_GBps": step_bytes / max(stats.device_secs, 1e-9) / 1e9},
        },
        # how much of a game's life the window covers (steady state needs the warm-up to be longer than a game)
        "window": {"warmup_batch_steps": warm_done, "timed_batch_steps": int(tot["done"]) * args.batch_steps,
                   "games_finished": int(tot["games"]), "positions_finished": int(tot["positions"])},
        "timing": {"session_open_s": t_open, "timed_s": elapsed},
    }
    if args.record:
        out["record"] = {"what": "every finished game's record built on the host (sink callback) and written as bundles to a "
                                 "tmpfs inside the timed region", "games_in_callback": rec_games[0], "bundle_files": rec_files}
    # Extra leg (N=1, network evaluators): the same workload with the games as ONE group, i.e. the gather kernel launched
    # over all resident games at once instead of two half-size launches pipelined against the evaluator. It is the
    # kernel's best per-launch figure; the timed configuration above trades it for throughput.
    if has_net and world == 1 and not args.no_full_launch and "AR_GROUPS" not in os.environ and \
            (time.perf_counter() - T_START) < args.deadline - 120.0:
        prof = under_profiler()
        if prof:
            out["roofline"]["full_launch"] = {"skipped": "profiler", "seen": prof}
        else:
            try:
                from alpharat_amd.sampling import release_device_memory

                release_device_memory(local_rank)  # (the arena block this process keeps cached between sessions)
                out["roofline"]["full_launch"] = full_launch_leg(args.evaluator, args.resident, args.batch_steps, local_rank)
            except Exception as e:  # noqa: BLE001 -- the extra leg never costs the headline line
                out["roofline"]["full_launch"] = {"error": str(e)}
    if not args.no_cpu_baseline and args.cpu_secs > 0 and world == 1:  # a reported baseline, timed on rank 0 at N=1 only
        left = args.deadline + 60.0 - (time.perf_counter() - T_START)
        if left > args.cpu_secs + 15.0:
            out["cpu_baseline"] = cpu_baseline(weights, args.evaluator, args.cpu_secs)
            if args.cpu_sweep:  # SURVEY 8d / bench_selfplay.rs:213-224: 1, 8, 32, 64, all threads (bounded samples)
                cores = os.cpu_count() or 1
                out["cpu_baseline"]["sweep"] = [
                    {k: v for k, v in cpu_baseline(weights, args.evaluator, min(args.cpu_secs, 8.0), threads=t).items()
                     if k in ("value", "wall_rate", "cores", "games_per_sec")}
                    for t in sorted({1, 8, 32, 64, cores}) if t <= cores]
        else:
            out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def _sum_windows(a, b):
    """Consecutive windows of one session: counters and times add up."""
    from alpharat_amd.sampling import SelfPlayStats

    kw = {k: getattr(a, k) + getattr(b, k) for k in SelfPlayStats._RAW}
    kw["min_turns"] = min(x for x in (a.min_turns, b.min_turns) if x) if (a.min_turns or b.min_turns) else 0
    kw["max_turns"] = max(a.max_turns, b.max_turns)
    return SelfPlayStats(**kw)


def _progress(rank: int, line: dict) -> None:
    if rank != 0:
        return
    s = json.dumps(line)
    print(s, file=sys.stderr, flush=True)
    try:
        d = ROOT / "gpurun_out"
        d.mkdir(exist_ok=True)
        with open(d / "bench_progress.jsonl", "a") as f:
            f.write(s + "\n")
    except OSError:
        pass


if __name__ == "__main__":
    sys.exit(main())
