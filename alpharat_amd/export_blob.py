"""Stand-in for the reference's ``scripts.export_onnx`` module on machines without the ``onnx`` package.

``run_rust_sampling`` calls ``_ensure_onnx(checkpoint)`` (``alpharat/data/rust_sampling.py:118-134``), which
imports ``export_onnx`` from ``scripts.export_onnx`` when no ``foo.onnx`` sits next to ``foo.pt``. The HIP sampler
never reads an ONNX graph: it reads the weight blob ``foo.arnet``. ``install()`` (called by the
``alpharat_sampling`` import-name shim) registers this module under the name ``scripts.export_onnx`` when ``onnx``
cannot be imported, so the unchanged reference code ends up writing the blob and hands
``rust_self_play(onnx_model_path="foo.onnx")`` a path that ``alpharat_amd.sampling`` maps to ``foo.arnet``.
With ``onnx`` installed nothing is replaced: the reference exports its graph as usual and the blob is made from
the ``.pt`` next to it on first use.
"""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path


def export_onnx(checkpoint_path, output_path=None, opset_version: int = 17, verify: bool = False) -> Path:
    """Same signature as ``scripts/export_onnx.py:55`` ``export_onnx``; writes ``<checkpoint>.arnet`` (cached by
    mtime) and returns the ``.onnx`` path the caller asked for (that file is not created)."""
    from .weights import checkpoint_to_blob

    pt = Path(checkpoint_path)
    checkpoint_to_blob(pt)
    return Path(output_path) if output_path is not None else pt.with_suffix(".onnx")


def install(force: bool = False) -> bool:
    """Register this module as ``scripts.export_onnx`` unless the real exporter can work (``onnx`` importable)."""
    if not force and importlib.util.find_spec("onnx") is not None:
        return False
    sys.modules["scripts.export_onnx"] = sys.modules[__name__]
    return True
