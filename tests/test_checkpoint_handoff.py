"""Checkpoint -> weight blob on real `.pt` files: the checkpoints under tests/golden/ckpt/
were saved by tools/gen_ckpt_golden.py in the trainer's layout (alpharat/nn/training/loop.py:392-424) from the
reference's own model / config classes, one of them with the `_orig_mod.` prefix of a compiled module.
CPU part: blob contents, caching next to the checkpoint (the reference caches `.onnx` the same way,
alpharat/data/rust_sampling.py:118-134), error behaviour (alpharat/config/checkpoint.py:59-82), and the zero-edit
route for `_ensure_onnx`. GPU part: the HIP network built from the blob reproduces the saved model's `predict()`."""
import os
import shutil
import sys
import time
from pathlib import Path

import numpy as np
import pytest

CKPT = Path(__file__).parent / "golden" / "ckpt"
NAMES = ["mlp_5x5_h32", "mlp_5x5_h32_compiled", "symmetric_5x5_h32", "cnn_gpool_7x5_c16"]


def _state_dict(pt):
    import torch

    c = torch.load(pt, map_location="cpu", weights_only=True)
    return c, {k.removeprefix("_orig_mod."): v.numpy() for k, v in c["model_state_dict"].items()}


@pytest.mark.parametrize("name", NAMES)
def test_blob_holds_the_checkpoints_state_dict(name, tmp_path):
    from alpharat_amd.weights import checkpoint_to_blob, read_blob

    pt = tmp_path / f"{name}.pt"
    shutil.copy(CKPT / f"{name}.pt", pt)
    blob = checkpoint_to_blob(pt)
    assert blob == pt.with_suffix(".arnet") and blob.exists()
    ckpt, sd = _state_dict(pt)
    arch, w, h, tensors = read_blob(blob)
    assert arch == ckpt["config"]["model"]["architecture"] and (w, h) == (ckpt["width"], ckpt["height"])
    floats = {k: v for k, v in sd.items() if np.issubdtype(v.dtype, np.floating)}
    assert sorted(tensors) == sorted(floats) and not any(k.startswith("_orig_mod.") for k in tensors)
    for k, v in floats.items():
        assert tensors[k].dtype == np.float32 and tensors[k].shape == v.shape
        assert tensors[k].tobytes() == np.ascontiguousarray(v, np.float32).tobytes(), k
    assert any("running_var" in k for k in tensors)  # BatchNorm statistics travel, counters do not
    assert not any("num_batches_tracked" in k for k in tensors)


def test_blob_is_cached_by_mtime_and_rebuilt_when_the_checkpoint_changes(tmp_path):
    from alpharat_amd.weights import checkpoint_to_blob, read_blob

    pt = tmp_path / "best_model.pt"
    shutil.copy(CKPT / "mlp_5x5_h32.pt", pt)
    blob = checkpoint_to_blob(pt)
    stamp = blob.stat().st_mtime_ns
    assert checkpoint_to_blob(pt) == blob and blob.stat().st_mtime_ns == stamp  # reused, not rewritten
    # a newer checkpoint at the same path (the trainer overwrites best_model.pt): the blob follows
    shutil.copy(CKPT / "symmetric_5x5_h32.pt", pt)
    future = time.time() + 5
    os.utime(pt, (future, future))
    assert read_blob(checkpoint_to_blob(pt))[0] == "symmetric"
    # an explicit destination
    other = checkpoint_to_blob(pt, tmp_path / "elsewhere.arnet")
    assert other.name == "elsewhere.arnet" and read_blob(other)[0] == "symmetric"


def test_checkpoints_without_size_or_architecture_are_refused(tmp_path):
    import torch

    from alpharat_amd.weights import checkpoint_to_blob

    c, _ = _state_dict(CKPT / "mlp_5x5_h32.pt")
    no_size = {k: v for k, v in c.items() if k not in ("width", "height")}
    torch.save(no_size, tmp_path / "a.pt")
    with pytest.raises(ValueError, match="width/height"):
        checkpoint_to_blob(tmp_path / "a.pt")
    no_arch = dict(c, config={"model": {"hidden_dim": 32}})
    torch.save(no_arch, tmp_path / "b.pt")
    with pytest.raises(ValueError, match="architecture"):
        checkpoint_to_blob(tmp_path / "b.pt")
    katago = dict(c, config={"model": {"architecture": "cnn_katago"}})
    torch.save(katago, tmp_path / "c.pt")
    with pytest.raises(ValueError, match="unsupported architecture"):
        checkpoint_to_blob(tmp_path / "c.pt")


def test_ensure_onnx_route_needs_no_reference_edit(tmp_path, monkeypatch):
    """What `_ensure_onnx` does (rust_sampling.py:118-134), step by step, against a stand-in `scripts` package:
    no foo.onnx -> `from scripts.export_onnx import export_onnx` -> export -> path handed to rust_self_play."""
    from alpharat_amd import export_blob
    from alpharat_amd.sampling import _resolve_weights

    pkg = tmp_path / "ref"
    (pkg / "scripts").mkdir(parents=True)
    (pkg / "scripts" / "__init__.py").write_text("")
    (pkg / "scripts" / "export_onnx.py").write_text("import onnx  # the real exporter needs a package that is absent\n")
    monkeypatch.syspath_prepend(str(pkg))
    for m in ("scripts", "scripts.export_onnx"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    assert export_blob.install(force=True)
    from scripts.export_onnx import export_onnx  # resolves to the stand-in, `import onnx` never runs

    pt = tmp_path / "best_model.pt"
    shutil.copy(CKPT / "cnn_gpool_7x5_c16.pt", pt)
    onnx_path = pt.with_suffix(".onnx")
    assert not onnx_path.exists()
    result = export_onnx(pt, onnx_path)
    assert Path(result) == onnx_path and not onnx_path.exists() and pt.with_suffix(".arnet").exists()
    # ... and rust_self_play(onnx_model_path=str(result)) picks the blob
    assert _resolve_weights(None, str(result)) == str(pt.with_suffix(".arnet"))
    # a second sampling run: `_ensure_onnx` exports again (no .onnx file), which is a cache hit
    stamp = pt.with_suffix(".arnet").stat().st_mtime_ns
    export_onnx(pt, onnx_path)
    assert pt.with_suffix(".arnet").stat().st_mtime_ns == stamp
    monkeypatch.delitem(sys.modules, "scripts.export_onnx", raising=False)


def test_missing_blob_and_checkpoint_is_an_error(tmp_path):
    from alpharat_amd.sampling import _resolve_weights

    with pytest.raises(RuntimeError, match="no weight blob"):
        _resolve_weights(None, str(tmp_path / "nothing.onnx"))
    assert _resolve_weights("given.arnet", str(tmp_path / "nothing.onnx")) == "given.arnet"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_net_from_checkpoint_reproduces_predict(name, tmp_path):
    from test_gpu_nets import _game_from_obs

    from alpharat_amd.nets import Net

    pt = tmp_path / f"{name}.pt"
    shutil.copy(CKPT / f"{name}.pt", pt)
    gold = np.load(CKPT / f"{name}.npz")
    w, h = (7, 5) if "7x5" in name else (5, 5)
    out = Net.from_checkpoint(pt).evaluate([_game_from_obs(o, w, h) for o in gold["obs"]])
    for k in ("logits_p1", "logits_p2", "policy_p1", "policy_p2", "value_p1", "value_p2"):
        np.testing.assert_allclose(out[k], gold[k], atol=1e-5, rtol=1e-5, err_msg=f"{name}:{k}")
