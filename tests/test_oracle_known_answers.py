"""Pins the CPU oracle: the reference's own exact-answer unit tests (restated in
oracle/test_oracle.cpp, each citing the reference test) and the 7 encoder golden vectors."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

import _oracle as O

GOLDEN = Path(__file__).parent / "golden" / "encoder"


def test_reference_unit_test_known_answers():
    O.lib()  # builds liboracle.so and test_oracle if stale
    exe = O.ORACLE_DIR / "test_oracle"
    if not exe.exists():
        subprocess.run(["make", "-s", "-C", str(O.ORACLE_DIR), "test_oracle"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


def game_from_fixture(fx) -> O.Game:
    xy = lambda d: (d["x"], d["y"])
    g = O.Game(
        fx["width"], fx["height"], fx["max_turns"], p1=xy(fx["p1_pos"]), p2=xy(fx["p2_pos"]),
        cheese=[xy(c) for c in fx["cheese"]],
        walls=[(xy(w["pos1"]), xy(w["pos2"])) for w in fx["walls"]],
        mud=[(xy(m["pos1"]), xy(m["pos2"]), m["value"]) for m in fx["mud"]],
    )
    for d1, d2 in fx.get("moves", []):
        g.make_move(d1, d2)
    return g


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("*.json")), ids=lambda p: p.stem)
def test_encoder_golden(path):
    # crates/alpharat-sampling/tests/parity.rs:16 -- tolerance 1e-6
    fx = json.loads(path.read_text())
    got = game_from_fixture(fx).encode()
    want = np.asarray(fx["expected"], dtype=np.float32)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=1e-6, rtol=0)


def test_fixture_engine_states():
    # SURVEY.md Appendix B rows pinned by the fixtures
    fx = json.loads((GOLDEN / "mud_stuck_5x5.json").read_text())
    s = game_from_fixture(fx).state()
    assert s["p1"] == (2, 0) and s["p1_mud"] == 3 and s["turn"] == 1
    fx = json.loads((GOLDEN / "midgame_5x5.json").read_text())
    s = game_from_fixture(fx).state()
    assert (s["p1_score"], s["p2_score"], s["turn"], s["remaining"]) == (1.0, 1.0, 1, 0)
    fx = json.loads((GOLDEN / "wall_5x5.json").read_text())
    g = game_from_fixture(fx)
    assert g.state()["p1_score"] == 0.0 and g.cheese_mask()[0] == 1  # start-cell cheese not taken at creation
