"""ctypes binding of libalpharat_hip.so (include/alpharat_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950). There is no CPU
fallback: if the library is missing, or no HIP device is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

PKG = Path(__file__).resolve().parent
# AR_LIB: another build of the same library (A/B measurements of two builds); never a different backend
LIB_PATH = Path(os.environ["AR_LIB"]) if os.environ.get("AR_LIB") else PKG / "libalpharat_hip.so"

AR_OK, AR_E_INVALID, AR_E_BACKEND, AR_E_IO, AR_E_DEVICE, AR_E_NOMEM = 0, -1, -2, -3, -4, -5


class ArSearchConfig(C.Structure):
    _fields_ = [
        ("c_puct", C.c_float), ("fpu_reduction", C.c_float), ("force_k", C.c_float),
        ("noise_epsilon", C.c_float), ("noise_concentration", C.c_float),
        ("collision_limit_min", C.c_uint32), ("collision_limit_max", C.c_uint32),
        ("collision_scaling_start", C.c_uint32), ("collision_scaling_end", C.c_uint32),
        ("collision_scaling_power", C.c_float),
    ]


class ArGameSpec(C.Structure):
    _fields_ = [
        ("width", C.c_uint8), ("height", C.c_uint8), ("max_turns", C.c_uint16), ("turn", C.c_uint16),
        ("p1_x", C.c_uint8), ("p1_y", C.c_uint8), ("p2_x", C.c_uint8), ("p2_y", C.c_uint8),
        ("p1_mud", C.c_uint8), ("p2_mud", C.c_uint8), ("p1_score", C.c_float), ("p2_score", C.c_float),
        ("cost", C.c_void_p), ("cheese", C.c_void_p),
    ]


class ArSearchResult(C.Structure):
    _fields_ = [
        ("policy_p1", C.c_float * 5), ("policy_p2", C.c_float * 5), ("value_p1", C.c_float), ("value_p2", C.c_float),
        ("visit_counts_p1", C.c_float * 5), ("visit_counts_p2", C.c_float * 5),
        ("prior_p1", C.c_float * 5), ("prior_p2", C.c_float * 5),
        ("total_visits", C.c_uint32), ("nn_evals", C.c_uint32), ("terminals", C.c_uint32), ("collisions", C.c_uint32),
    ]


class ArLeaf(C.Structure):
    _fields_ = [
        ("p1_x", C.c_uint8), ("p1_y", C.c_uint8), ("p2_x", C.c_uint8), ("p2_y", C.c_uint8),
        ("p1_mud", C.c_uint8), ("p2_mud", C.c_uint8), ("turn", C.c_uint16),
        ("p1_score", C.c_float), ("p2_score", C.c_float), ("cheese_bits", C.c_uint64 * 4),
    ]


ArPredictFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ArLeaf), C.c_uint32, C.POINTER(C.c_float),
                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float))


class ArSelfPlayParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint8), ("height", C.c_uint8), ("cheese_count", C.c_uint16), ("max_turns", C.c_uint16),
        ("num_games", C.c_uint32), ("cheese_symmetric", C.c_int), ("maze_type", C.c_char_p), ("positions", C.c_char_p),
        ("wall_density", C.c_float), ("mud_density", C.c_float), ("maze_symmetric", C.c_int),
        ("simulations", C.c_uint32), ("batch_size", C.c_uint32), ("search", ArSearchConfig),
        ("num_threads", C.c_uint32), ("output_dir", C.c_char_p), ("max_games_per_bundle", C.c_uint32),
        ("weights_path", C.c_char_p), ("device", C.c_char_p), ("mux_max_batch_size", C.c_uint32),
        ("cache_size", C.c_uint64), ("has_seed", C.c_int), ("game_seed_base", C.c_uint64),
        ("rng_seed_base", C.c_uint64), ("first_game_index", C.c_uint32), ("concurrent_games", C.c_uint32),
        ("device_index", C.c_int),
    ]


class ArSelfPlayStats(C.Structure):
    _fields_ = [
        ("total_games", C.c_uint32), ("total_positions", C.c_uint64), ("total_simulations", C.c_uint64),
        ("elapsed_secs", C.c_double), ("p1_wins", C.c_uint32), ("p2_wins", C.c_uint32), ("draws", C.c_uint32),
        ("total_cheese_collected", C.c_float), ("total_cheese_available", C.c_uint32),
        ("min_turns", C.c_uint32), ("max_turns", C.c_uint32),
        ("total_nn_evals", C.c_uint64), ("total_terminals", C.c_uint64), ("total_collisions", C.c_uint64),
        ("cache_hits", C.c_uint64), ("cache_misses", C.c_uint64),
        ("gather_node_visits", C.c_uint64), ("backup_node_visits", C.c_uint64), ("new_nodes", C.c_uint64),
        ("device_secs", C.c_double), ("steps", C.c_uint64),
        ("gather_secs", C.c_double), ("gather_launches", C.c_uint64),
    ]


class ArSessionInfo(C.Structure):
    _fields_ = [
        ("resident_games", C.c_uint32), ("groups", C.c_uint32), ("gather_kind", C.c_uint32), ("gather_pass_limit", C.c_uint32),
        ("tree_region_bytes", C.c_uint64), ("host_grown_arenas", C.c_uint64), ("idle_slots", C.c_uint32),
        ("tree_pages_per_game", C.c_float),
    ]


class ArProgress(C.Structure):
    _fields_ = [
        ("games_completed", C.c_uint32), ("positions_completed", C.c_uint64),
        ("simulations_completed", C.c_uint64), ("nn_evals_completed", C.c_uint64),
    ]


class ArGameRecordView(C.Structure):
    _fields_ = [
        ("width", C.c_uint8), ("height", C.c_uint8), ("max_turns", C.c_uint16),
        ("game_index", C.c_uint32), ("n_positions", C.c_uint32),
        ("maze", C.POINTER(C.c_int8)), ("initial_cheese", C.POINTER(C.c_uint8)), ("cheese_outcomes", C.POINTER(C.c_uint8)),
        ("final_p1_score", C.c_float), ("final_p2_score", C.c_float), ("result", C.c_uint8),
        ("cheese_available", C.c_uint16),
        ("total_simulations", C.c_uint64), ("total_nn_evals", C.c_uint64), ("total_terminals", C.c_uint64),
        ("total_collisions", C.c_uint64),
        ("p1_pos", C.POINTER(C.c_uint8)), ("p2_pos", C.POINTER(C.c_uint8)),
        ("p1_score", C.POINTER(C.c_float)), ("p2_score", C.POINTER(C.c_float)),
        ("p1_mud", C.POINTER(C.c_uint8)), ("p2_mud", C.POINTER(C.c_uint8)),
        ("turn", C.POINTER(C.c_uint16)), ("cheese_mask", C.POINTER(C.c_uint8)),
        ("value_p1", C.POINTER(C.c_float)), ("value_p2", C.POINTER(C.c_float)),
        ("visit_counts_p1", C.POINTER(C.c_float)), ("visit_counts_p2", C.POINTER(C.c_float)),
        ("prior_p1", C.POINTER(C.c_float)), ("prior_p2", C.POINTER(C.c_float)),
        ("policy_p1", C.POINTER(C.c_float)), ("policy_p2", C.POINTER(C.c_float)),
        ("action_p1", C.POINTER(C.c_uint8)), ("action_p2", C.POINTER(C.c_uint8)),
    ]


ArGameSink = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(ArGameRecordView))

# every symbol include/alpharat_hip.h declares (checked by the CPU test-suite)
EXPORTS = {
    "ar_version": (C.c_char_p, []),
    "ar_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "ar_device_count": (C.c_int, []),
    "ar_generate_maze": (C.c_int, [C.c_uint8, C.c_uint8, C.c_float, C.c_float, C.c_int, C.c_uint64, C.c_void_p]),
    "ar_generate_cheese": (C.c_int, [C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int, C.c_uint64,
                                    C.c_void_p]),
    "ar_device_sync": (C.c_int, [C.c_int]),
    "ar_release_device_memory": (C.c_int, [C.c_int]),
    "ar_net_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ar_net_free": (None, [C.c_void_p]),
    "ar_net_evaluate": (C.c_int, [C.c_void_p, C.POINTER(ArGameSpec), C.c_uint32] + [C.c_void_p] * 6),
    "ar_encode": (C.c_int, [C.POINTER(ArGameSpec), C.c_uint32, C.c_int, C.c_void_p]),
    "ar_search": (C.c_int, [C.POINTER(ArGameSpec), C.POINTER(ArSearchConfig), C.c_uint32, C.c_uint32,
                            C.POINTER(C.c_uint64), ArPredictFn, C.c_void_p, C.c_void_p, C.c_int,
                            C.POINTER(ArSearchResult)]),
    "ar_search_many": (C.c_int, [C.POINTER(ArGameSpec), C.c_uint32, C.POINTER(ArSearchConfig), C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint64), C.c_void_p, C.c_int, C.POINTER(ArSearchResult)]),
    "ar_selfplay_run": (C.c_int, [C.POINTER(ArSelfPlayParams), C.POINTER(ArProgress), ArGameSink, C.c_void_p,
                                  C.POINTER(ArSelfPlayStats)]),
    "ar_selfplay_open": (C.c_int, [C.POINTER(ArSelfPlayParams), C.POINTER(ArProgress), ArGameSink, C.c_void_p,
                                   C.POINTER(C.c_void_p)]),
    "ar_selfplay_step": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(ArSelfPlayStats), C.POINTER(C.c_int)]),
    "ar_selfplay_close": (C.c_int, [C.c_void_p, C.POINTER(ArSelfPlayStats)]),
    "ar_selfplay_info": (C.c_int, [C.c_void_p, C.POINTER(ArSessionInfo)]),
    "ar_write_bundle": (C.c_int, [C.POINTER(ArGameRecordView), C.c_uint32, C.c_char_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library. Raises if it has not been built -- there is no other backend."""
    global _lib
    if _lib is not None:
        return _lib
    # one hardware queue per stream of the step pipeline (see alpharat_hip.hip "hardware queues"): effective when the
    # HIP runtime has not been initialised in this process yet
    # -- so it is only set when nothing can have brought the runtime up before (torch with an initialised device has)
    import sys

    torch = sys.modules.get("torch")
    runtime_up = bool(torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized())
    if not runtime_up:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    elif "GPU_MAX_HW_QUEUES" not in os.environ:
        # nobody asked for more queues before the runtime came up: it has its default four, and setting the variable now
        # would only make the library believe otherwise -- it then keeps one group of games (no extra streams)
        os.environ.setdefault("AR_HW_QUEUES_UNKNOWN", "1")
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). alpharat_amd has no CPU fallback."
        )
    L = C.CDLL(str(LIB_PATH))
    for name, (res, args) in EXPORTS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    buf = C.create_string_buffer(4096)
    load().ar_last_error(buf, 4096)
    return buf.value.decode(errors="replace")


def check(rc: int) -> None:
    """Map C-ABI error codes onto the exceptions the reference's PyO3 layer raises
    (crates/alpharat-sampling/src/bindings.rs:474-482, crates/alpharat-mcts/src/bindings.rs:300-303)."""
    if rc == AR_OK:
        return
    msg = last_error()
    if rc == AR_E_INVALID:
        raise ValueError(msg)
    if rc == AR_E_IO:
        raise OSError(msg)
    if rc == AR_E_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)
