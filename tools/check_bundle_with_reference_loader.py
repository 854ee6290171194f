"""One-off check, in the build container only (needs /root/reference): a bundle written by ar_write_bundle
is read back by the reference's own loader (alpharat/data/loader.py load_game_bundle) and gives the
known answers of tests/data/test_rust_bundle_parity.py. The reference is imported, never copied; shims
for the modules this image lacks come from tools/gen_net_golden.py."""
from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))

import gen_net_golden  # noqa: E402

gen_net_golden._install_shims()
sys.path.insert(0, "/root/reference")
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
from alpharat.data.loader import is_bundle_file, load_game_bundle  # noqa: E402

import test_bundle_known_answer as T  # noqa: E402
from alpharat_amd import _lib  # noqa: E402

lib = _lib.load()
keep: list = []
views = (_lib.ArGameRecordView * 2)(T._view(T._game0(), keep), T._view(T._game1(), keep))
with tempfile.TemporaryDirectory() as d:
    path = Path(d) / "bundle_check.npz"
    lib.ar_write_bundle.restype = C.c_int
    _lib.check(lib.ar_write_bundle(views, 2, str(path).encode()))
    assert is_bundle_file(path)
    games = load_game_bundle(path)
    assert len(games) == 2
    g0, g1 = games
    assert (g0.width, g0.height, g0.max_turns, g0.result) == (3, 3, 30, 0)
    assert abs(g0.final_p1_score - 0.5) < 1e-6 and abs(g0.final_p2_score - 0.5) < 1e-6
    assert g0.maze.shape == (3, 3, 4) and g0.maze[0, 0, 2] == -1 and g0.maze[0, 0, 0] == 1
    assert g0.initial_cheese[1, 1] and g0.initial_cheese.sum() == 1
    assert len(g0.positions) == 2 and g0.positions[0].p1_pos == (0, 0) and g0.positions[0].p2_pos == (2, 2)
    np.testing.assert_allclose(g0.positions[0].policy_p1, [0.625, 0.3125, 0.0, 0.0, 0.0625], atol=1e-6)
    np.testing.assert_allclose(g0.positions[0].visit_counts_p2, [0.0, 0.0, 6.0, 8.0, 2.0], atol=1e-6)
    assert g0.positions[1].action_p1 == 1 and g0.positions[1].action_p2 == 3
    assert (1, 1) in g0.positions[0].cheese_positions
    assert (g1.max_turns, g1.result) == (20, 1) and len(g1.positions) == 1
    assert g1.positions[0].p2_mud == 3 and g1.positions[0].action_p2 == 4
    np.testing.assert_allclose(g1.positions[0].policy_p2, [0, 0, 0, 0, 1], atol=1e-6)
print("reference loader reads the bundle: ok")
