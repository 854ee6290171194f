"""Drop-in for crates/alpharat-mcts-python/python/alpharat_sampling/__init__.py (MI355X backend)."""
from alpharat_amd.sampling import (
    SelfPlayProgress,
    SelfPlayStats,
    preload_cuda_libs,
    preload_tensorrt_libs,
    rust_self_play,
)

# NN-guided sampling without editing the reference: see alpharat_amd/export_blob.py
from alpharat_amd import export_blob as _export_blob

_export_blob.install()

__all__ = ["SelfPlayStats", "SelfPlayProgress", "rust_self_play", "preload_cuda_libs", "preload_tensorrt_libs"]
