"""ctypes driver for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"


class OrSearchConfig(C.Structure):
    _fields_ = [
        ("c_puct", C.c_float),
        ("fpu_reduction", C.c_float),
        ("force_k", C.c_float),
        ("noise_epsilon", C.c_float),
        ("noise_concentration", C.c_float),
        ("collision_limit_min", C.c_uint32),
        ("collision_limit_max", C.c_uint32),
        ("collision_scaling_start", C.c_uint32),
        ("collision_scaling_end", C.c_uint32),
        ("collision_scaling_power", C.c_float),
    ]


class OrSearchResult(C.Structure):
    _fields_ = [
        ("policy_p1", C.c_float * 5),
        ("policy_p2", C.c_float * 5),
        ("value_p1", C.c_float),
        ("value_p2", C.c_float),
        ("visit_counts_p1", C.c_float * 5),
        ("visit_counts_p2", C.c_float * 5),
        ("prior_p1", C.c_float * 5),
        ("prior_p2", C.c_float * 5),
        ("total_visits", C.c_uint32),
        ("nn_evals", C.c_uint32),
        ("terminals", C.c_uint32),
        ("collisions", C.c_uint32),
    ]


def make_config(**kw) -> OrSearchConfig:
    d = dict(
        c_puct=1.5,
        fpu_reduction=0.2,
        force_k=2.0,
        noise_epsilon=0.0,
        noise_concentration=10.83,
        collision_limit_min=1,
        collision_limit_max=256,
        collision_scaling_start=800,
        collision_scaling_end=50000,
        collision_scaling_power=1.0,
    )
    d.update(kw)
    return OrSearchConfig(**d)


_lib = None


def build() -> None:
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    so = ORACLE_DIR / "liboracle.so"
    srcs = list(ORACLE_DIR.glob("*.hpp")) + list(ORACLE_DIR.glob("*.cpp")) + list(ORACLE_DIR.glob("*.inc"))
    if not so.exists() or (srcs and max(s.stat().st_mtime for s in srcs) > so.stat().st_mtime):
        if os.access(ORACLE_DIR, os.W_OK):
            build()
    L = C.CDLL(str(so))
    vp, u8, u16, u32, u64, i32 = C.c_void_p, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "or_last_error": (C.c_char_p, []),
        "or_game_new": (vp, [u8, u8, u16]),
        "or_game_clone": (vp, [vp]),
        "or_game_free": (None, [vp]),
        "or_game_set_positions": (None, [vp, u8, u8, u8, u8]),
        "or_game_add_wall": (i32, [vp, i32, i32, i32, i32]),
        "or_game_add_mud": (i32, [vp, i32, i32, i32, i32, i32]),
        "or_game_add_cheese": (None, [vp, i32, i32]),
        "or_game_random_cheese": (i32, [vp, u16, i32, u64]),
        "or_game_random_maze": (None, [vp, C.c_float, C.c_float, i32, u64]),
        "or_game_cost": (None, [vp, vp]),
        "or_game_make_move": (None, [vp, u8, u8]),
        "or_game_over": (i32, [vp]),
        "or_game_state": (None, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
        "or_game_cheese_mask": (None, [vp, vp]),
        "or_game_maze": (None, [vp, vp]),
        "or_game_effective_actions": (None, [vp, vp, vp]),
        "or_encode": (None, [vp, vp]),
        "or_net_load": (vp, [C.c_char_p]),
        "or_net_free": (None, [vp]),
        "or_net_forward": (i32, [vp, vp, i32, i32, vp]),
        "or_tree_new": (vp, [vp]),
        "or_tree_free": (None, [vp]),
        "or_tree_advance": (i32, [vp, vp, u8, u8]),
        "or_tree_node_count": (u32, [vp]),
        "or_tree_check": (None, [vp, vp]),
        "or_tree_dump": (u32, [vp, vp, u32]),
        "or_rng_seed": (None, [u64, vp]),
        "or_rng_next_u64": (u64, [vp]),
        "or_rng_gen_range": (u32, [vp, u32]),
        "or_rng_weighted5": (i32, [vp, vp]),
        "or_rng_gamma": (C.c_double, [vp, C.c_double]),
        "or_rng_normal": (C.c_double, [vp]),
        "or_search": (
            i32,
            [vp, vp, C.POINTER(OrSearchConfig), u32, u32, vp, i32, C.c_float, C.c_float, vp, C.POINTER(OrSearchResult), vp],
        ),
        "or_play_game": (vp, [vp, C.POINTER(OrSearchConfig), u32, u32, u64, i32, vp, u32]),
        "or_record_free": (None, [vp]),
        "or_record_header": (None, [vp, vp, vp, vp]),
        "or_record_game_arrays": (None, [vp, vp, vp, vp]),
        "or_record_positions": (None, [vp, vp, vp, vp]),
        "or_selfplay_bench": (
            i32,
            [u8, u8, u16, u16, u32, C.POINTER(OrSearchConfig), u32, u32, u32, u64, u64, i32, vp, vp, C.POINTER(C.c_double),
             C.c_double, vp, vp],
        ),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Game:
    """Oracle game handle. Coordinates are (x, y), y up."""

    def __init__(self, w, h, max_turns=100, p1=None, p2=None, cheese=(), walls=(), mud=(), _handle=None):
        L = lib()
        self.w, self.h = w, h
        if _handle is not None:
            self.g = _handle
            return
        self.g = L.or_game_new(w, h, max_turns)
        p1 = p1 if p1 is not None else (0, 0)
        p2 = p2 if p2 is not None else (w - 1, h - 1)
        L.or_game_set_positions(self.g, p1[0], p1[1], p2[0], p2[1])
        for (a, b) in walls:
            assert L.or_game_add_wall(self.g, a[0], a[1], b[0], b[1])
        for (a, b, v) in mud:
            assert L.or_game_add_mud(self.g, a[0], a[1], b[0], b[1], v)
        for c in cheese:
            L.or_game_add_cheese(self.g, c[0], c[1])

    def __del__(self):
        try:
            lib().or_game_free(self.g)
        except Exception:
            pass

    def clone(self) -> "Game":
        return Game(self.w, self.h, _handle=lib().or_game_clone(self.g))

    def random_cheese(self, count, symmetric=True, seed=0):
        assert lib().or_game_random_cheese(self.g, count, int(symmetric), seed)
        return self

    def random_maze(self, wall_density=0.7, mud_density=0.1, symmetric=True, seed=0):
        """Own generator (the engine's is absent): oracle/pyrat_engine.hpp make_maze. Call before random_cheese."""
        lib().or_game_random_maze(self.g, wall_density, mud_density, int(symmetric), seed)
        return self

    def cost(self) -> np.ndarray:
        out = np.zeros((self.h, self.w, 4), np.uint8)
        lib().or_game_cost(self.g, _ptr(out))
        return out

    def make_move(self, d1, d2):
        lib().or_game_make_move(self.g, d1, d2)

    def over(self) -> bool:
        return bool(lib().or_game_over(self.g))

    def state(self):
        i = (C.c_int32 * 8)()
        f = (C.c_float * 2)()
        lib().or_game_state(self.g, i, f)
        return dict(
            p1=(i[0], i[1]), p2=(i[2], i[3]), p1_mud=i[4], p2_mud=i[5], turn=i[6], remaining=i[7],
            p1_score=f[0], p2_score=f[1],
        )

    def cheese_mask(self) -> np.ndarray:
        m = np.zeros(self.w * self.h, dtype=np.uint8)
        lib().or_game_cheese_mask(self.g, _ptr(m))
        return m

    def maze(self) -> np.ndarray:
        m = np.zeros(self.w * self.h * 4, dtype=np.int8)
        lib().or_game_maze(self.g, _ptr(m))
        return m.reshape(self.h, self.w, 4)

    def effective_actions(self):
        a = np.zeros(5, dtype=np.uint8)
        b = np.zeros(5, dtype=np.uint8)
        lib().or_game_effective_actions(self.g, _ptr(a), _ptr(b))
        return a.tolist(), b.tolist()

    def encode(self) -> np.ndarray:
        o = np.zeros(self.w * self.h * 7 + 6, dtype=np.float32)
        lib().or_encode(self.g, _ptr(o))
        return o


class Rng:
    def __init__(self, seed):
        self.s = np.zeros(4, dtype=np.uint64)
        lib().or_rng_seed(seed, _ptr(self.s))

    def next_u64(self):
        return lib().or_rng_next_u64(_ptr(self.s))

    def gen_range(self, n):
        return lib().or_rng_gen_range(_ptr(self.s), n)

    def weighted5(self, w):
        w = np.asarray(w, dtype=np.float32)
        return lib().or_rng_weighted5(_ptr(self.s), _ptr(w))

    def gamma(self, shape):
        return lib().or_rng_gamma(_ptr(self.s), shape)

    def normal(self):
        return lib().or_rng_normal(_ptr(self.s))


class Net:
    def __init__(self, path):
        self.n = lib().or_net_load(str(path).encode())
        if not self.n:
            raise RuntimeError(lib().or_last_error().decode())

    def __del__(self):
        try:
            lib().or_net_free(self.n)
        except Exception:
            pass

    def forward(self, obs: np.ndarray) -> dict:
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        n, d = obs.shape
        out = np.zeros((n, 22), dtype=np.float32)
        if lib().or_net_forward(self.n, _ptr(obs), n, d, _ptr(out)) != 0:
            raise RuntimeError(lib().or_last_error().decode())
        return dict(
            logits_p1=out[:, 0:5], logits_p2=out[:, 5:10], policy_p1=out[:, 10:15], policy_p2=out[:, 15:20],
            value_p1=out[:, 20], value_p2=out[:, 21],
        )


def result_to_dict(r: OrSearchResult) -> dict:
    d = {}
    for k, _ in OrSearchResult._fields_:
        v = getattr(r, k)
        d[k] = np.array(list(v), dtype=np.float32) if hasattr(v, "__len__") else v
    return d


class Tree:
    def __init__(self, game: Game):
        self.t = lib().or_tree_new(game.g)

    def __del__(self):
        try:
            lib().or_tree_free(self.t)
        except Exception:
            pass

    def search(self, game: Game, cfg: OrSearchConfig, n_sims, batch, rng: Rng, backend=0, v1=0.0, v2=0.0, net: Net | None = None):
        res = OrSearchResult()
        ctr = np.zeros(3, dtype=np.uint64)
        rc = lib().or_search(
            self.t, game.g, C.byref(cfg), n_sims, batch, _ptr(rng.s), backend, v1, v2,
            net.n if net is not None else None, C.byref(res), _ptr(ctr),
        )
        if rc != 0:
            raise RuntimeError(lib().or_last_error().decode())
        d = result_to_dict(res)
        d["counters"] = ctr
        return d

    def advance(self, game_after: Game, a1, a2) -> bool:
        return bool(lib().or_tree_advance(self.t, game_after.g, a1, a2))

    def node_count(self) -> int:
        return lib().or_tree_node_count(self.t)

    def check(self) -> dict:
        o = np.zeros(4, dtype=np.uint64)
        lib().or_tree_check(self.t, _ptr(o))
        return dict(nodes=int(o[0]), in_flight=int(o[1]), bad_edge_sums=int(o[2]), terminal_with_children=int(o[3]))

    def dump(self, max_nodes=1 << 20) -> np.ndarray:
        n = self.node_count()
        n = min(max(n, 1), max_nodes)
        out = np.zeros((n, 43), dtype=np.uint32)
        cnt = lib().or_tree_dump(self.t, _ptr(out), n)
        return out[: min(cnt, n)]


def search_once(game: Game, cfg=None, n_sims=100, batch=8, seed=42, **kw):
    cfg = cfg or make_config()
    t = Tree(game)
    r = t.search(game, cfg, n_sims, batch, Rng(seed), **kw)
    r["tree"] = t
    return r


def play_game(game: Game, cfg: OrSearchConfig, n_sims, batch, rng_seed, backend=0, net: Net | None = None, game_index=0) -> dict:
    L = lib()
    h = L.or_play_game(game.g, C.byref(cfg), n_sims, batch, rng_seed, backend, net.n if net else None, game_index)
    if not h:
        raise RuntimeError(L.or_last_error().decode())
    try:
        hdr = np.zeros(7, dtype=np.int32)
        sums = np.zeros(7, dtype=np.uint64)
        fs = np.zeros(2, dtype=np.float32)
        L.or_record_header(h, _ptr(hdr), _ptr(sums), _ptr(fs))
        n, w, hh = int(hdr[0]), int(hdr[1]), int(hdr[2])
        hw = w * hh
        maze = np.zeros(hw * 4, dtype=np.int8)
        ic = np.zeros(hw, dtype=np.uint8)
        co = np.zeros(hw, dtype=np.uint8)
        L.or_record_game_arrays(h, _ptr(maze), _ptr(ic), _ptr(co))
        ints = np.zeros((max(n, 1), 9), dtype=np.int32)
        fl = np.zeros((max(n, 1), 34), dtype=np.float32)
        masks = np.zeros((max(n, 1), hw), dtype=np.uint8)
        L.or_record_positions(h, _ptr(ints), _ptr(fl), _ptr(masks))
        return dict(
            n=n, width=w, height=hh, max_turns=int(hdr[3]), result=int(hdr[4]), cheese_available=int(hdr[5]),
            game_index=int(hdr[6]), total_simulations=int(sums[0]), total_nn_evals=int(sums[1]),
            total_terminals=int(sums[2]), total_collisions=int(sums[3]), gather_node_visits=int(sums[4]),
            backup_node_visits=int(sums[5]), new_nodes=int(sums[6]), final_p1_score=float(fs[0]),
            final_p2_score=float(fs[1]), maze=maze.reshape(hh, w, 4), initial_cheese=ic.reshape(hh, w),
            cheese_outcomes=co.reshape(hh, w), ints=ints[:n], floats=fl[:n], masks=masks[:n],
        )
    finally:
        L.or_record_free(h)


def selfplay_bench(w, h, cheese, max_turns, n_games, cfg, n_sims, batch, threads, game_seed_base=0,
                   rng_seed_base=0xA1FA0000, backend=0, net: Net | None = None, max_secs=0.0) -> dict:
    """`max_secs` > 0: no thread claims a new game after that long. `thread_rate_sum` adds up the per-thread
    rates (simulations of a thread / the time its last game ended)."""
    out = np.zeros(9, dtype=np.uint64)
    el = C.c_double(0)
    tsecs = np.zeros(threads, dtype=np.float64)
    tsims = np.zeros(threads, dtype=np.uint64)
    rc = lib().or_selfplay_bench(w, h, cheese, max_turns, n_games, C.byref(cfg), n_sims, batch, threads,
                                 game_seed_base, rng_seed_base, backend, net.n if net else None, _ptr(out), C.byref(el),
                                 float(max_secs), _ptr(tsecs), _ptr(tsims))
    if rc != 0:
        raise RuntimeError("oracle selfplay bench failed")
    keys = ["games", "positions", "simulations", "nn_evals", "terminals", "collisions", "gather_node_visits",
            "backup_node_visits", "new_nodes"]
    d = {k: int(v) for k, v in zip(keys, out)}
    d["elapsed_secs"] = el.value
    ok = tsecs > 0
    d["thread_rate_sum"] = float((tsims[ok] / tsecs[ok]).sum()) if ok.any() else 0.0
    return d


class OrLeaf(C.Structure):
    _fields_ = [("p1x", C.c_uint8), ("p1y", C.c_uint8), ("p2x", C.c_uint8), ("p2y", C.c_uint8), ("p1_mud", C.c_uint8),
                ("p2_mud", C.c_uint8), ("turn", C.c_uint16), ("p1_score", C.c_float), ("p2_score", C.c_float),
                ("cheese", C.c_uint32 * 8)]


OrEvalFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(OrLeaf), C.c_uint32, C.POINTER(C.c_float))


class OrCallback(C.Structure):
    _fields_ = [("fn", OrEvalFn), ("user", C.c_void_p)]


class CallbackBackend:
    """Backend kind 4: `evaluate(leaves: list[dict]) -> (p1[n,5], p2[n,5], v1[n], v2[n])` called by the oracle's
    search for every leaf batch. A leaf dict has p1, p2 (x, y), p1_mud, p2_mud, turn, p1_score, p2_score and
    `cheese` (uint8[256] mask, idx = y*w + x). Pass as `net=` with `backend=4`."""

    def __init__(self, evaluate):
        self.calls = 0
        self.sizes = []

        def _fn(_user, leaves, n, out):
            try:
                ls = []
                for i in range(n):
                    l = leaves[i]
                    bits = np.frombuffer(bytes(l.cheese), dtype=np.uint8)
                    ls.append(dict(p1=(l.p1x, l.p1y), p2=(l.p2x, l.p2y), p1_mud=l.p1_mud, p2_mud=l.p2_mud, turn=l.turn,
                                   p1_score=l.p1_score, p2_score=l.p2_score,
                                   cheese=np.unpackbits(bits, bitorder="little")))
                p1, p2, v1, v2 = evaluate(ls)
                res = np.concatenate([np.asarray(p1, np.float32).reshape(n, 5), np.asarray(p2, np.float32).reshape(n, 5),
                                      np.asarray(v1, np.float32).reshape(n, 1), np.asarray(v2, np.float32).reshape(n, 1)],
                                     axis=1)
                C.memmove(out, res.ctypes.data, n * 48)
                self.calls += 1
                self.sizes.append(n)
                return 0
            except Exception:  # noqa: BLE001 -- reported to the oracle as an evaluator failure
                import traceback

                traceback.print_exc()
                return 1

        self._fn = OrEvalFn(_fn)
        self._cb = OrCallback(self._fn, None)
        self.n = C.addressof(self._cb)
