#!/usr/bin/env python3
"""Generate `.pt` checkpoints in the layout the reference's trainer saves
(`alpharat/nn/training/loop.py:392-424`: epoch, model_state_dict, optimizer_state_dict, val_loss,
best_val_loss, config{model, optim, data, game}, width, height) by importing the reference's own model and
config classes (build container only -- /root/reference does not travel; the files are committed under
tests/golden/ckpt/ with the `predict()` outputs of the very model that was saved).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_ckpt_golden.py

One of them is saved from a `torch.compile`-style wrapper, i.e. with the `_orig_mod.` key prefix
(`alpharat/config/checkpoint.py:24-104` loads those too). The checkpoint-to-blob hand-off of the sampling path
(`alpharat_amd.weights.checkpoint_to_blob`, replacing `_ensure_onnx`, `alpharat/data/rust_sampling.py:118-134`)
is tested on these files.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


def main() -> int:
    import gen_net_golden as G

    G._install_shims()
    sys.path.insert(0, str(G.REF))
    import numpy as np
    import torch

    from alpharat.nn.architectures.cnn.config import CNNModelConfig, CNNOptimConfig
    from alpharat.nn.architectures.mlp.config import MLPModelConfig, MLPOptimConfig
    from alpharat.nn.architectures.symmetric.config import SymmetricModelConfig, SymmetricOptimConfig
    from alpharat.nn.training.keys import ModelOutput

    out_dir = ROOT / "tests" / "golden" / "ckpt"
    out_dir.mkdir(parents=True, exist_ok=True)
    cases = [
        ("mlp_5x5_h32", MLPModelConfig(hidden_dim=32), MLPOptimConfig(), (5, 5), False),
        ("mlp_5x5_h32_compiled", MLPModelConfig(hidden_dim=32, dropout=0.1), MLPOptimConfig(), (5, 5), True),
        ("symmetric_5x5_h32", SymmetricModelConfig(hidden_dim=32), SymmetricOptimConfig(), (5, 5), False),
        ("cnn_gpool_7x5_c16", CNNModelConfig(trunk=dict(channels=16, blocks=[dict(type="res"), dict(type="gpool", gpool_channels=8)]),
                                              player_dim=8, hidden_dim=16), CNNOptimConfig(), (7, 5), False),
    ]
    for idx, (name, mc, oc, (w, h), compiled) in enumerate(cases):
        torch.manual_seed(5000 + idx)
        gen = torch.Generator().manual_seed(6000 + idx)
        mc.set_data_dimensions(w, h)
        model = mc.build_model()
        G.randomise_bn(model, gen)
        model.eval()
        obs = G.synth_obs(np.random.default_rng(7000 + idx), w, h, 16, max_turns=50)
        with torch.no_grad():
            pred = model.predict(torch.from_numpy(obs))
            fwd = model.forward(torch.from_numpy(obs))
        sd = model.state_dict()
        if compiled:  # what state_dict() of a torch.compile'd module looks like
            sd = {f"_orig_mod.{k}": v for k, v in sd.items()}
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        torch.save({
            "epoch": 3,
            "model_state_dict": sd,
            "optimizer_state_dict": opt.state_dict(),
            "val_loss": 1.25,
            "best_val_loss": 1.25,
            "config": {"model": mc.model_dump(), "optim": oc.model_dump(), "data": {"train_dir": "x", "val_dir": "y"},
                       "game": None},
            "width": w,
            "height": h,
        }, out_dir / f"{name}.pt")
        np.savez_compressed(
            out_dir / f"{name}.npz", obs=obs,
            policy_p1=pred[ModelOutput.POLICY_P1].numpy(), policy_p2=pred[ModelOutput.POLICY_P2].numpy(),
            value_p1=pred[ModelOutput.VALUE_P1].numpy(), value_p2=pred[ModelOutput.VALUE_P2].numpy(),
            logits_p1=fwd[ModelOutput.LOGITS_P1].numpy(), logits_p2=fwd[ModelOutput.LOGITS_P2].numpy())
        print(f"{name}: {(out_dir / f'{name}.pt').stat().st_size} B, architecture {mc.architecture}")
    return 0


if __name__ == "__main__":
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.exit(main())
