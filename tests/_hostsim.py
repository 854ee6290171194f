"""ctypes driver for tests/hostsim/libhostsim.so: the product's device-side search headers
compiled for the CPU (test harness only, see tests/hostsim/hostsim.cpp)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "hostsim"


class HsCfg(C.Structure):
    _fields_ = [
        ("c_puct", C.c_float), ("fpu_reduction", C.c_float), ("force_k", C.c_float),
        ("noise_epsilon", C.c_float), ("noise_concentration", C.c_float),
        ("coll_min", C.c_uint32), ("coll_max", C.c_uint32), ("coll_start", C.c_uint32), ("coll_end", C.c_uint32),
        ("coll_power", C.c_float),
    ]


class HsGame(C.Structure):
    _fields_ = [
        ("width", C.c_uint8), ("height", C.c_uint8), ("max_turns", C.c_uint16), ("turn", C.c_uint16),
        ("p1_x", C.c_uint8), ("p1_y", C.c_uint8), ("p2_x", C.c_uint8), ("p2_y", C.c_uint8),
        ("p1_mud", C.c_uint8), ("p2_mud", C.c_uint8), ("p1_score", C.c_float), ("p2_score", C.c_float),
        ("cost", C.c_void_p), ("cheese", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
        L = C.CDLL(str(HERE / "libhostsim.so"))
        L.hs_run.restype = C.c_void_p
        L.hs_run.argtypes = [C.POINTER(HsGame), C.POINTER(HsCfg), C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int,
                             C.c_float, C.c_float, C.c_uint32]
        L.hs_free.argtypes = [C.c_void_p]
        for n in ("hs_header", "hs_final", "hs_positions", "hs_last"):
            getattr(L, n).restype = None
        L.hs_header.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_final.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_tree_dump.restype = C.c_uint32
        L.hs_tree_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


def cfg_from_oracle(ocfg) -> HsCfg:
    return HsCfg(ocfg.c_puct, ocfg.fpu_reduction, ocfg.force_k, ocfg.noise_epsilon, ocfg.noise_concentration,
                 ocfg.collision_limit_min, ocfg.collision_limit_max, ocfg.collision_scaling_start,
                 ocfg.collision_scaling_end, ocfg.collision_scaling_power)


def game_arrays(og):
    """(cost[hw*4] u8, cheese[hw] u8, state dict) from an oracle Game (tests/_oracle.py)."""
    maze = og.maze().reshape(-1).astype(np.int16)
    cost = np.where(maze < 0, 0, maze).astype(np.uint8)
    return cost, og.cheese_mask().astype(np.uint8), og.state()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(og, max_turns, ocfg, n_sims, batch, seed, single=False, eval_mode=0, v1=0.0, v2=0.0, arena_nodes=0) -> dict:
    """Play a whole game (or one search when single=True) from the oracle game `og`'s position."""
    L = lib()
    cost, cheese, st = game_arrays(og)
    g = HsGame(og.w, og.h, max_turns, st["turn"], st["p1"][0], st["p1"][1], st["p2"][0], st["p2"][1], st["p1_mud"],
               st["p2_mud"], st["p1_score"], st["p2_score"], _p(cost), _p(cheese))
    cfg = cfg_from_oracle(ocfg)
    h = L.hs_run(C.byref(g), C.byref(cfg), n_sims, batch, seed, int(single), eval_mode, v1, v2, arena_nodes)
    try:
        hdr = np.zeros(14, dtype=np.uint64)
        fs = np.zeros(2, dtype=np.float32)
        L.hs_header(h, _p(hdr), _p(fs))
        n = int(hdr[0])
        hw = og.w * og.h
        ints = np.zeros((max(n, 1), 9), dtype=np.int32)
        fl = np.zeros((max(n, 1), 34), dtype=np.float32)
        masks = np.zeros((max(n, 1), hw), dtype=np.uint8)
        L.hs_positions(h, _p(ints), _p(fl), _p(masks))
        fin = np.zeros(6, dtype=np.int32)
        fmask = np.zeros(hw, dtype=np.uint8)
        L.hs_final(h, _p(fin), _p(fmask))
        last = np.zeros(34, dtype=np.float32)
        cnt = np.zeros(4, dtype=np.uint32)
        L.hs_last(h, _p(last), _p(cnt))
        nodes = int(hdr[4])
        dump = np.zeros((max(nodes, 1), 43), dtype=np.uint32)
        got = L.hs_tree_dump(h, _p(dump), max(nodes, 1))
        return dict(
            n=n, status=int(hdr[1]), error=int(hdr[2]), grows=int(hdr[3]), node_count=nodes,
            total_simulations=int(hdr[5]), total_nn_evals=int(hdr[6]), total_terminals=int(hdr[7]),
            total_collisions=int(hdr[8]), gather_node_visits=int(hdr[9]), backup_node_visits=int(hdr[10]),
            new_nodes=int(hdr[11]), wide_passes=int(hdr[12]), wide_gathers=int(hdr[13]), final_p1_score=float(fs[0]), final_p2_score=float(fs[1]),
            ints=ints[:n], floats=fl[:n], masks=masks[:n], final=fin, final_mask=fmask, last=last, last_counts=cnt,
            dump=dump[: min(got, max(nodes, 1))], dump_count=got,
        )
    finally:
        L.hs_free(h)


def hashed_eval(width):
    """The CPU harness's stand-in network (hostsim.cpp hashed_eval) as an `evaluate` function for
    _oracle.CallbackBackend: priors and values that are a fixed hash of the position."""
    M = 0xFFFFFFFF

    def evaluate(leaves):
        n = len(leaves)
        p = np.zeros((2, n, 5), dtype=np.float32)
        v = np.zeros((2, n), dtype=np.float32)
        for i, l in enumerate(leaves):
            c1 = l["p1"][1] * width + l["p1"][0]
            c2 = l["p2"][1] * width + l["p2"][0]
            x = (c1 * 7919 + c2 * 104729 + l["turn"] * 1299709) & M
            for c in np.flatnonzero(l["cheese"]):
                x = (x + (int(c) + 1) * 15485863) & M
            for pl in range(2):
                w = [np.float32(1 + ((((x + a * 40503 + pl * 7) & M) * 2654435761 & M) >> 8) % 1000) for a in range(5)]
                tot = np.float32(0.0)
                for a in range(5):
                    tot = np.float32(tot + w[a])
                for a in range(5):
                    p[pl, i, a] = np.float32(w[a] / tot)
            v[0, i] = np.float32(((x * 2246822519 & M) >> 10) % 64) / np.float32(16.0)
            v[1, i] = np.float32(((x * 3266489917 & M) >> 10) % 64) / np.float32(16.0)
        return p[0], p[1], v[0], v[1]

    return evaluate
