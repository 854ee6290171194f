#!/usr/bin/env python3
"""Generate golden input/output vectors for the policy/value heads by importing the
reference's own Python model classes (build container only -- /root/reference does not
travel to the GPU box; the vectors are committed under tests/golden/nets/).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_net_golden.py

For each case: a seeded random state_dict (BatchNorm running stats and affine params are
randomised too, so eval-mode BN is not an identity), N observations in the reference's flat
layout, and the outputs of `model.predict()` (policies, values) and `model.forward()` (logits).
The state_dict is stored in the repo's weight-blob format (alpharat_amd/weights.py).
"""
from __future__ import annotations

import enum
import os
import sys
import types
import typing
from pathlib import Path

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))


def _install_shims() -> None:
    """py3.11-only names and absent packages the reference imports at module load."""
    import datetime
    import importlib.machinery

    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)
        enum.StrEnum = StrEnum
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc

    def stub(name: str, **attrs) -> types.ModuleType:
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Dir(enum.IntEnum):
        UP = 0
        RIGHT = 1
        DOWN = 2
        LEFT = 3
        STAY = 4

    for name in ("hydra", "hydra.core", "hydra.core.global_hydra", "omegaconf", "tensorboard",
                 "torch.utils.tensorboard", "pyrat_engine", "pyrat_engine.core", "pyrat_engine.core.game",
                 "pyrat_engine.core.builder"):
        if name not in sys.modules:
            stub(name)
    stub("pyrat_engine.core.types", Direction=_Dir, Coordinates=object, Wall=object, Mud=object)
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.modules["hydra.core.global_hydra"].GlobalHydra = object
    sys.modules["hydra"].compose = None
    sys.modules["hydra"].initialize_config_dir = None
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["omegaconf"].DictConfig = dict


def synth_obs(rng, w: int, h: int, n: int, max_turns: int):
    """Valid flat observations of an open maze (alpharat/nn/builders/flat.py:142-197 layout)."""
    import numpy as np

    hw = w * h
    obs = np.zeros((n, hw * 7 + 6), dtype=np.float32)
    maze = np.full((h, w, 4), 0.1, dtype=np.float32)
    maze[h - 1, :, 0] = -1
    maze[:, w - 1, 1] = -1
    maze[0, :, 2] = -1
    maze[:, 0, 3] = -1
    for i in range(n):
        o = obs[i]
        o[: hw * 4] = maze.reshape(-1)
        p1, p2 = rng.integers(0, hw, size=2)
        o[hw * 4 + p1] = 1.0
        o[hw * 5 + p2] = 1.0
        k = int(rng.integers(1, max(2, hw // 4)))
        cheese = rng.choice(hw, size=k, replace=False)
        o[hw * 6 + cheese] = 1.0
        s1 = float(rng.integers(0, 9)) * 0.5
        s2 = float(rng.integers(0, 9)) * 0.5
        turn = int(rng.integers(0, max_turns))
        m1 = int(rng.integers(0, 4)) if i % 5 == 0 else 0
        m2 = int(rng.integers(0, 4)) if i % 7 == 0 else 0
        o[hw * 7:] = [s1 - s2, np.float32(turn) / np.float32(max_turns), np.float32(m1) / np.float32(10),
                      np.float32(m2) / np.float32(10), np.float32(s1) / np.float32(10), np.float32(s2) / np.float32(10)]
    return obs


def randomise_bn(model, gen) -> None:
    import torch

    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) * 1.5 + 0.25)
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
        if isinstance(m, torch.nn.Linear) and m.bias is not None:
            with torch.no_grad():
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
    # heads are initialised at std 0.01 (near-uniform); widen them so logits are informative
    for name, p in model.named_parameters():
        if ("policy" in name or "value_head" in name) and name.endswith("weight"):
            with torch.no_grad():
                p.mul_(6.0)


def main() -> int:
    _install_shims()
    sys.path.insert(0, str(REF))
    import numpy as np
    import torch

    from alpharat.nn.models.mlp import PyRatMLP
    from alpharat.nn.models.symmetric import SymmetricMLP
    from alpharat.nn.architectures.cnn.config import CNNModelConfig
    from alpharat.nn.training.keys import ModelOutput
    from alpharat_amd.weights import write_blob

    out_dir = ROOT / "tests" / "golden" / "nets"
    out_dir.mkdir(parents=True, exist_ok=True)
    cases = [
        # name, arch, (w,h), kwargs
        ("mlp_5x5_h32", "mlp", (5, 5), dict(hidden_dim=32)),
        ("mlp_7x7_h256", "mlp", (7, 7), dict(hidden_dim=256)),
        ("symmetric_5x5_h32", "symmetric", (5, 5), dict(hidden_dim=32)),
        ("symmetric_7x7_h256", "symmetric", (7, 7), dict(hidden_dim=256)),
        ("cnn_res_5x5_c16", "cnn", (5, 5), dict(trunk=dict(channels=16, blocks=[dict(type="res")]))),
        ("cnn_gpool_7x7_c64", "cnn", (7, 7), dict(trunk=dict(channels=64, blocks=[
            dict(type="res"), dict(type="res"), dict(type="gpool", gpool_channels=32)]),
            player_dim=32, hidden_dim=64, dropout=0.1)),
        ("cnn_gpool_7x5_c16", "cnn", (7, 5), dict(trunk=dict(channels=16, blocks=[
            dict(type="gpool", gpool_channels=8), dict(type="res")]), player_dim=8, hidden_dim=16)),
        # the value head that pools the trunk's output (value_head.type "pooled", cnn/heads.py:40-68)
        ("cnn_pooled_7x5_c16", "cnn", (7, 5), dict(trunk=dict(channels=16, blocks=[
            dict(type="res"), dict(type="gpool", gpool_channels=8)]), value_head=dict(type="pooled"), player_dim=8,
            hidden_dim=16)),
        ("cnn_pooled_7x7_c32", "cnn", (7, 7), dict(trunk=dict(channels=32, blocks=[dict(type="res"), dict(type="res")]),
                                                   value_head=dict(type="pooled"), player_dim=16, hidden_dim=32)),
        # boards above 8x8 (configs/game/15x11_open_asymmetric.yaml; cnn/model.py:167-230 is size-agnostic): one leaf over
        # two row tiles per wavefront on the matrix cores, and the FMA path's row / column loops
        ("cnn_gpool_15x11_c32", "cnn", (15, 11), dict(trunk=dict(channels=32, blocks=[
            dict(type="res"), dict(type="gpool", gpool_channels=16)]), value_head=dict(type="pooled"), player_dim=16,
            hidden_dim=32)),
        ("cnn_gpool_9x10_c16", "cnn", (9, 10), dict(trunk=dict(channels=16, blocks=[
            dict(type="gpool", gpool_channels=8), dict(type="res")]), player_dim=8, hidden_dim=16)),
        ("cnn_res_15x11_c64", "cnn", (15, 11), dict(trunk=dict(channels=64, blocks=[dict(type="res")]), player_dim=16,
                                                  hidden_dim=32)),
    ]
    only = set(sys.argv[1:])  # (names on the command line: regenerate just those; seeds depend on the case's index only)
    for idx, (name, arch, (w, h), kw) in enumerate(cases):
        if only and name not in only:
            continue
        torch.manual_seed(1000 + idx)
        gen = torch.Generator().manual_seed(2000 + idx)
        obs_dim = w * h * 7 + 6
        if arch == "mlp":
            model = PyRatMLP(obs_dim=obs_dim, **kw)
        elif arch == "symmetric":
            model = SymmetricMLP(width=w, height=h, **kw)
        else:
            cfg = CNNModelConfig(**kw)
            cfg.set_data_dimensions(w, h)
            model = cfg.build_model()
        randomise_bn(model, gen)
        model.eval()
        rng = np.random.default_rng(3000 + idx)
        obs = synth_obs(rng, w, h, 24, max_turns=50)
        with torch.no_grad():
            x = torch.from_numpy(obs)
            pred = model.predict(x)
            fwd = model.forward(x)
        blob = out_dir / f"{name}.arnet"
        write_blob(blob, arch, w, h, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
        np.savez_compressed(
            out_dir / f"{name}.npz",
            obs=obs,
            policy_p1=pred[ModelOutput.POLICY_P1].numpy(), policy_p2=pred[ModelOutput.POLICY_P2].numpy(),
            value_p1=pred[ModelOutput.VALUE_P1].numpy(), value_p2=pred[ModelOutput.VALUE_P2].numpy(),
            logits_p1=fwd[ModelOutput.LOGITS_P1].numpy(), logits_p2=fwd[ModelOutput.LOGITS_P2].numpy(),
        )
        n_params = sum(p.numel() for p in model.parameters())
        print(f"{name}: params={n_params} blob={blob.stat().st_size} B")
    return 0


if __name__ == "__main__":
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.exit(main())
