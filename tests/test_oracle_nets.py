"""Oracle policy/value heads vs golden vectors generated from the reference's Python classes
(tools/gen_net_golden.py). Tolerance: 1e-5 on logits / policies / values (north star)."""
from pathlib import Path

import numpy as np
import pytest

import _oracle as O

NETS = Path(__file__).parent / "golden" / "nets"
CASES = sorted(p.stem for p in NETS.glob("*.arnet"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_net_matches_reference_outputs(name):
    gold = np.load(NETS / f"{name}.npz")
    net = O.Net(NETS / f"{name}.arnet")
    out = net.forward(gold["obs"])
    for k in ("logits_p1", "logits_p2", "policy_p1", "policy_p2", "value_p1", "value_p2"):
        np.testing.assert_allclose(out[k], gold[k], atol=1e-5, rtol=1e-5, err_msg=f"{name}:{k}")


def test_blob_roundtrip(tmp_path):
    from alpharat_amd.weights import read_blob, write_blob

    t = {"a.weight": np.arange(12, dtype=np.float32).reshape(3, 4), "a.num_batches_tracked": np.array(3)}
    p = write_blob(tmp_path / "x.arnet", "mlp", 5, 7, t)
    arch, w, h, back = read_blob(p)
    assert (arch, w, h) == ("mlp", 5, 7) and list(back) == ["a.weight"]
    np.testing.assert_array_equal(back["a.weight"], t["a.weight"])
