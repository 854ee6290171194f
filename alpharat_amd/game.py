"""A PyRat-compatible game object for the Python side of the drop-in boundary.

The reference passes ``pyrat_engine.core.game.PyRat`` objects (Rust-backed, third party) into
``rust_mcts_search`` and hands lists of them to ``predict_fn``. The HIP sampler only needs the
attribute surface documented in
``crates/alpharat-mcts-python/python/pyrat_engine/core/game.pyi`` -- any object with that
surface works (``spec_from_game`` duck-types it). This class provides the same surface without
the Rust engine, for callers and tests that do not have ``pyrat_engine`` installed, and for the
leaf objects given to ``predict_fn``.

Rules restated here are the ones pinned by the reference; the device and
the oracle implement the same rules independently.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, NamedTuple

import numpy as np

UP, RIGHT, DOWN, LEFT, STAY = 0, 1, 2, 3, 4
_DELTA = {UP: (0, 1), RIGHT: (1, 0), DOWN: (0, -1), LEFT: (-1, 0)}


class Coordinates(NamedTuple):
    x: int
    y: int


@dataclass(frozen=True)
class Wall:
    pos1: Coordinates
    pos2: Coordinates


@dataclass(frozen=True)
class Mud:
    pos1: Coordinates
    pos2: Coordinates
    value: int


@dataclass
class MoveUndo:
    p1_pos: Coordinates
    p2_pos: Coordinates
    p1_mud: int
    p2_mud: int
    p1_score: float
    p2_score: float
    collected_cheese: list
    turn: int


def _xy(p) -> Coordinates:
    if hasattr(p, "x"):
        return Coordinates(int(p.x), int(p.y))
    return Coordinates(int(p[0]), int(p[1]))


class PyRat:
    """Simultaneous-move PyRat position (y up, origin bottom-left)."""

    def __init__(self, width: int, height: int, cost: np.ndarray, cheese: np.ndarray, p1, p2, max_turns: int = 300,
                 turn: int = 0, p1_score: float = 0.0, p2_score: float = 0.0, p1_mud: int = 0, p2_mud: int = 0,
                 total_cheese: int | None = None) -> None:
        self._w, self._h = int(width), int(height)
        self._cost = np.ascontiguousarray(cost, dtype=np.uint8).reshape(self._h * self._w * 4)
        self._cheese = np.ascontiguousarray(cheese, dtype=np.uint8).reshape(self._h * self._w).copy()
        self._p1, self._p2 = _xy(p1), _xy(p2)
        self._max_turns, self._turn = int(max_turns), int(turn)
        self._s1, self._s2 = float(p1_score), float(p2_score)
        self._m1, self._m2 = int(p1_mud), int(p2_mud)
        self._total = int(total_cheese) if total_cheese is not None else int(self._s1 + self._s2 + self._cheese.sum())

    # ---- construction (game.pyi:133-160 create_custom) ------------------------------------------
    @staticmethod
    def open_cost(width: int, height: int) -> np.ndarray:
        c = np.ones((height, width, 4), dtype=np.uint8)
        c[height - 1, :, UP] = 0
        c[:, width - 1, RIGHT] = 0
        c[0, :, DOWN] = 0
        c[:, 0, LEFT] = 0
        return c

    @staticmethod
    def create_custom(width: int, height: int, walls: Iterable = (), mud: Iterable = (), cheese: Iterable = (),
                      player1_pos=None, player2_pos=None, max_turns: int = 300, symmetric: bool = True) -> "PyRat":
        cost = PyRat.open_cost(width, height)

        def edge(a, b, v):
            a, b = _xy(a), _xy(b)
            for d, (dx, dy) in _DELTA.items():
                if (a.x + dx, a.y + dy) == (b.x, b.y):
                    cost[a.y, a.x, d] = v
                    cost[b.y, b.x, (d + 2) % 4] = v
                    return
            raise ValueError(f"cells {a} and {b} are not adjacent")

        for w_ in walls:
            a, b = (w_.pos1, w_.pos2) if hasattr(w_, "pos1") else w_
            edge(a, b, 0)
        for m in mud:
            a, b, v = (m.pos1, m.pos2, m.value) if hasattr(m, "pos1") else m
            edge(a, b, int(v))
        mask = np.zeros((height, width), dtype=np.uint8)
        for c in cheese:
            c = _xy(c)
            mask[c.y, c.x] = 1
        p1 = player1_pos if player1_pos is not None else (0, 0)
        p2 = player2_pos if player2_pos is not None else (width - 1, height - 1)
        return PyRat(width, height, cost, mask, p1, p2, max_turns)

    def clone(self) -> "PyRat":
        return PyRat(self._w, self._h, self._cost, self._cheese, self._p1, self._p2, self._max_turns, self._turn,
                     self._s1, self._s2, self._m1, self._m2, self._total)

    # ---- read-only surface (game.pyi:240-300) -----------------------------------------------------
    width = property(lambda s: s._w)
    height = property(lambda s: s._h)
    turn = property(lambda s: s._turn)
    max_turns = property(lambda s: s._max_turns)
    player1_position = property(lambda s: s._p1)
    player2_position = property(lambda s: s._p2)
    player1_score = property(lambda s: s._s1)
    player2_score = property(lambda s: s._s2)
    player1_mud_turns = property(lambda s: s._m1)
    player2_mud_turns = property(lambda s: s._m2)

    def cheese_positions(self) -> list[Coordinates]:
        ys, xs = np.where(self._cheese.reshape(self._h, self._w))
        return [Coordinates(int(x), int(y)) for x, y in zip(xs, ys)]

    def wall_entries(self) -> list[Wall]:
        out = []
        c = self._cost.reshape(self._h, self._w, 4)
        for y in range(self._h):
            for x in range(self._w):
                if x + 1 < self._w and c[y, x, RIGHT] == 0:
                    out.append(Wall(Coordinates(x, y), Coordinates(x + 1, y)))
                if y + 1 < self._h and c[y, x, UP] == 0:
                    out.append(Wall(Coordinates(x, y), Coordinates(x, y + 1)))
        return out

    def mud_entries(self) -> list[Mud]:
        out = []
        c = self._cost.reshape(self._h, self._w, 4)
        for y in range(self._h):
            for x in range(self._w):
                if x + 1 < self._w and c[y, x, RIGHT] >= 2:
                    out.append(Mud(Coordinates(x, y), Coordinates(x + 1, y), int(c[y, x, RIGHT])))
                if y + 1 < self._h and c[y, x, UP] >= 2:
                    out.append(Mud(Coordinates(x, y), Coordinates(x, y + 1), int(c[y, x, UP])))
        return out

    def effective_actions(self, pos) -> list[int]:
        p = _xy(pos)
        c = self._cost.reshape(self._h, self._w, 4)[p.y, p.x]
        return [a if c[a] else STAY for a in range(4)] + [STAY]

    def effective_actions_p1(self) -> list[int]:
        return [STAY] * 5 if self._m1 > 0 else self.effective_actions(self._p1)

    def effective_actions_p2(self) -> list[int]:
        return [STAY] * 5 if self._m2 > 0 else self.effective_actions(self._p2)

    def effective_moves(self, pos) -> list[int]:
        p = _xy(pos)
        c = self._cost.reshape(self._h, self._w, 4)[p.y, p.x]
        return [a for a in range(4) if c[a]]

    # ---- rules ------------------------------------------------------------------------------------
    def _move(self, pos: Coordinates, mud: int, d: int) -> tuple[Coordinates, int]:
        if mud > 0:
            return pos, mud - 1
        if d >= 4:
            return pos, 0
        c = int(self._cost.reshape(self._h, self._w, 4)[pos.y, pos.x, d])
        if c == 0:
            return pos, 0
        dx, dy = _DELTA[d]
        return Coordinates(pos.x + dx, pos.y + dy), (c if c >= 2 else 0)

    def make_move(self, p1_move: int, p2_move: int) -> MoveUndo:
        undo = MoveUndo(self._p1, self._p2, self._m1, self._m2, self._s1, self._s2, [], self._turn)
        self._p1, self._m1 = self._move(self._p1, self._m1, int(p1_move))
        self._p2, self._m2 = self._move(self._p2, self._m2, int(p2_move))
        ch = self._cheese.reshape(self._h, self._w)
        f1, f2 = self._m1 == 0, self._m2 == 0
        if f1 and f2 and self._p1 == self._p2:
            if ch[self._p1.y, self._p1.x]:
                ch[self._p1.y, self._p1.x] = 0
                self._s1 += 0.5
                self._s2 += 0.5
                undo.collected_cheese.append(self._p1)
        else:
            if f1 and ch[self._p1.y, self._p1.x]:
                ch[self._p1.y, self._p1.x] = 0
                self._s1 += 1.0
                undo.collected_cheese.append(self._p1)
            if f2 and ch[self._p2.y, self._p2.x]:
                ch[self._p2.y, self._p2.x] = 0
                self._s2 += 1.0
                undo.collected_cheese.append(self._p2)
        self._turn += 1
        return undo

    def unmake_move(self, undo: MoveUndo) -> None:
        ch = self._cheese.reshape(self._h, self._w)
        for c in undo.collected_cheese:
            ch[c.y, c.x] = 1
        self._p1, self._p2, self._m1, self._m2 = undo.p1_pos, undo.p2_pos, undo.p1_mud, undo.p2_mud
        self._s1, self._s2, self._turn = undo.p1_score, undo.p2_score, undo.turn

    def is_over(self) -> bool:
        if self._turn >= self._max_turns or int(self._cheese.sum()) == 0:
            return True
        return self._s1 > self._total / 2 or self._s2 > self._total / 2

    def step(self, p1_move: int, p2_move: int) -> tuple[bool, list[Coordinates]]:
        u = self.make_move(p1_move, p2_move)
        return self.is_over(), u.collected_cheese

    # ---- arrays for the C-ABI ---------------------------------------------------------------------
    def cost_array(self) -> np.ndarray:
        return self._cost

    def cheese_array(self) -> np.ndarray:
        return self._cheese


def arrays_from_game(game) -> tuple[np.ndarray, np.ndarray]:
    """(cost[h*w*4], cheese[h*w]) from any object with the PyRat attribute surface."""
    if isinstance(game, PyRat):
        return game.cost_array(), game.cheese_array()
    w, h = int(game.width), int(game.height)
    cost = PyRat.open_cost(w, h)
    for wl in game.wall_entries():
        a, b = _xy(wl.pos1), _xy(wl.pos2)
        for d, (dx, dy) in _DELTA.items():
            if (a.x + dx, a.y + dy) == (b.x, b.y):
                cost[a.y, a.x, d] = 0
                cost[b.y, b.x, (d + 2) % 4] = 0
    for m in game.mud_entries():
        a, b = _xy(m.pos1), _xy(m.pos2)
        for d, (dx, dy) in _DELTA.items():
            if (a.x + dx, a.y + dy) == (b.x, b.y):
                cost[a.y, a.x, d] = int(m.value)
                cost[b.y, b.x, (d + 2) % 4] = int(m.value)
    cheese = np.zeros((h, w), dtype=np.uint8)
    for c in game.cheese_positions():
        c = _xy(c)
        cheese[c.y, c.x] = 1
    return np.ascontiguousarray(cost.reshape(-1)), np.ascontiguousarray(cheese.reshape(-1))
