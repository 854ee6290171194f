"""Device-resident policy/value heads (``ar_net_*``): load a weight blob, evaluate positions.

Replaces the reference's ``OnnxBackend`` / ``TensorrtBackend`` objects
(crates/alpharat-sampling/src/backends/) on the Python side of the boundary.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Any

import numpy as np

from . import _lib
from .mcts import spec_from_game


class Net:
    def __init__(self, blob_path: str | Path, device: int = 0) -> None:
        L = _lib.load()
        h = C.c_void_p()
        _lib.check(L.ar_net_load(str(blob_path).encode(), device, C.byref(h)))
        self.handle = h

    @classmethod
    def from_checkpoint(cls, checkpoint: str | Path, device: int = 0) -> "Net":
        from .weights import checkpoint_to_blob

        return cls(checkpoint_to_blob(checkpoint), device)

    def close(self) -> None:
        if getattr(self, "handle", None):
            _lib.load().ar_net_free(self.handle)
            self.handle = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, games: list[Any]) -> dict[str, np.ndarray]:
        """``model.predict`` over positions: policies (softmax), values (softplus) and raw logits."""
        n = len(games)
        keep: list = []
        specs = (_lib.ArGameSpec * n)(*[spec_from_game(g, keep) for g in games])
        out = {k: np.zeros((n, 5), np.float32) for k in ("policy_p1", "policy_p2", "logits_p1", "logits_p2")}
        out["value_p1"] = np.zeros(n, np.float32)
        out["value_p2"] = np.zeros(n, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(_lib.load().ar_net_evaluate(self.handle, specs, n, p(out["policy_p1"]), p(out["policy_p2"]),
                                               p(out["value_p1"]), p(out["value_p2"]), p(out["logits_p1"]),
                                               p(out["logits_p2"])))
        return out


def encode(games: list[Any], device: int = 0) -> np.ndarray:
    """Flat observations from the device encoder (== FlatObservationBuilder.build)."""
    n = len(games)
    keep: list = []
    specs = (_lib.ArGameSpec * n)(*[spec_from_game(g, keep) for g in games])
    dim = int(games[0].width) * int(games[0].height) * 7 + 6
    obs = np.zeros((n, dim), np.float32)
    _lib.check(_lib.load().ar_encode(specs, n, device, obs.ctypes.data_as(C.c_void_p)))
    return obs
