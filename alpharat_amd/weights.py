"""Weight hand-off: torch ``state_dict`` / ``.pt`` checkpoint -> flat weight blob.

Replaces the reference's ONNX export step for the sampling path
(``alpharat/data/rust_sampling.py:118-134`` ``_ensure_onnx`` -> ``scripts/export_onnx.py:55``):
the HIP evaluators read this blob instead of an ONNX graph.

Blob layout ("ARNET001", little endian):
    char[8] magic, u32 arch (0 mlp, 1 symmetric, 2 cnn), u32 width, u32 height, u32 n_tensors,
    then per tensor: u32 name_len, name, u32 ndim, u32 dims[ndim], f32 data (C order).
Tensor names are the torch ``state_dict`` keys; integer buffers (num_batches_tracked) are dropped.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Mapping

import numpy as np

ARCH_IDS = {"mlp": 0, "symmetric": 1, "cnn": 2}
MAGIC = b"ARNET001"


def write_blob(path: str | Path, arch: str, width: int, height: int, tensors: Mapping[str, np.ndarray]) -> Path:
    if arch not in ARCH_IDS:
        raise ValueError(f"unsupported architecture {arch!r} (supported: {sorted(ARCH_IDS)})")
    path = Path(path)
    items = []
    for name, arr in tensors.items():
        a = np.asarray(arr)
        if not np.issubdtype(a.dtype, np.floating):
            continue  # num_batches_tracked
        items.append((name, np.ascontiguousarray(a, dtype="<f4")))
    tmp = path.with_suffix(path.suffix + ".tmp")
    with open(tmp, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<IIII", ARCH_IDS[arch], width, height, len(items)))
        for name, a in items:
            nb = name.encode()
            fh.write(struct.pack("<I", len(nb)))
            fh.write(nb)
            fh.write(struct.pack("<I", a.ndim))
            fh.write(struct.pack(f"<{a.ndim}I", *a.shape))
            fh.write(a.tobytes())
    tmp.replace(path)
    return path


def read_blob(path: str | Path) -> tuple[str, int, int, dict[str, np.ndarray]]:
    data = Path(path).read_bytes()
    if data[:8] != MAGIC:
        raise ValueError(f"{path}: not an ARNET001 weight blob")
    arch_id, width, height, n = struct.unpack_from("<IIII", data, 8)
    off = 24
    out: dict[str, np.ndarray] = {}
    for _ in range(n):
        (nl,) = struct.unpack_from("<I", data, off)
        off += 4
        name = data[off : off + nl].decode()
        off += nl
        (nd,) = struct.unpack_from("<I", data, off)
        off += 4
        dims = struct.unpack_from(f"<{nd}I", data, off)
        off += 4 * nd
        cnt = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(data, dtype="<f4", count=cnt, offset=off).reshape(dims).copy()
        off += 4 * cnt
    arch = {v: k for k, v in ARCH_IDS.items()}[arch_id]
    return arch, width, height, out


def checkpoint_to_blob(checkpoint_path: str | Path, blob_path: str | Path | None = None) -> Path:
    """``.pt`` checkpoint (``alpharat/nn/training/loop.py:392-424`` layout: ``model_state_dict``,
    ``config.model.architecture``, ``width``, ``height``) -> blob next to it (``.arnet``), cached
    like the reference caches ``.onnx`` next to the ``.pt``."""
    import torch

    pt = Path(checkpoint_path)
    blob = Path(blob_path) if blob_path is not None else pt.with_suffix(".arnet")
    if blob.exists() and blob.stat().st_mtime >= pt.stat().st_mtime:
        return blob
    ckpt = torch.load(pt, map_location="cpu", weights_only=True)
    width, height = ckpt.get("width"), ckpt.get("height")
    if width is None or height is None:
        raise ValueError(f"Checkpoint {pt} missing width/height.")
    arch = (ckpt.get("config", {}).get("model", {}) or {}).get("architecture")
    if arch is None:
        raise ValueError(f"Checkpoint {pt} missing config.model.architecture.")
    sd = {k.removeprefix("_orig_mod."): v.detach().cpu().numpy() for k, v in ckpt["model_state_dict"].items()}
    return write_blob(blob, arch, int(width), int(height), sd)
