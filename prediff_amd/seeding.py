"""Deterministic, key-addressed random weights and inputs.

No checkpoint can be downloaded offline and the reference's default init makes the denoiser output identically zero
(final_proj is zero-initialised, cuboid_transformer_unet.py:342 -> apply_initialization mode "2"), so the
benchmarks, the smoke test, the golden-vector generator (tests/golden/gen_golden.py) and the parity tests all regenerate
every floating-point tensor of a ``state_dict`` from (seed, key) alone.  Nothing here touches the compute path.
"""
import math
import zlib
from typing import Dict, Mapping

import torch


def seeded_tensor(key: str, shape, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    shape = tuple(shape)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return r * (1.0 / math.sqrt(fan_in))
    if key.endswith(".weight"):          # 1-D weights are norm scales
        return 1.0 + 0.1 * r
    return 0.1 * r                        # biases


def seeded_state_dict(template: Mapping[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """Same keys/shapes as `template`; floating tensors regenerated, integer buffers kept."""
    out = {}
    for k in template:
        v = template[k]
        if torch.is_floating_point(v):
            out[k] = seeded_tensor(k, v.shape, seed)
        else:
            out[k] = v.detach().clone().cpu()
    return out


def heavy_tailed_state_dict(template: Mapping[str, torch.Tensor], seed: int, nu: float = 3.0, outlier: float = 30.0,
                            n_outlier: int = 2) -> Dict[str, torch.Tensor]:
    """Checkpoint-like stress weights (no trained checkpoint exists offline): every matrix / filter is Student-t(nu) distributed with
    the fan-in-scaled variance of `seeded_state_dict` (nu = 3: a few entries per tensor sit at 10-20 sigma), and `n_outlier` output
    channels of every matrix / filter and `n_outlier` entries of every norm scale are `outlier` times larger -- the outlier-channel
    pattern trained transformers show in their LayerNorm gains and the projections behind them.  Used by the robustness parity tests
    of the 16-bit / 8-bit engines (overflow of the half-precision packers, the fixed e4m3 activation scale)."""
    out = {}
    for k in template:
        v = template[k]
        if not torch.is_floating_point(v):
            out[k] = v.detach().clone().cpu()
            continue
        shape = tuple(v.shape)
        g = torch.Generator(device="cpu")
        g.manual_seed((seed * 1000003 + zlib.crc32(("ht:" + k).encode())) % (2 ** 63 - 1))
        if len(shape) >= 2:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            z = torch.randn(shape, generator=g, dtype=torch.float32)
            # Student-t(nu) = z / sqrt(chi2(nu) / nu); chi2(nu) as a sum of nu squared normals (nu integer), seeded by the same generator
            c2 = torch.zeros(shape)
            for _ in range(int(nu)):
                c2 += torch.randn(shape, generator=g, dtype=torch.float32) ** 2
            t = z / torch.sqrt(c2 / nu)
            t = t * math.sqrt((nu - 2.0) / nu) * (1.0 / math.sqrt(fan_in))        # unit-variance t, then the fan-in scale
            rows = torch.randperm(shape[0], generator=g)[:min(n_outlier, shape[0])]
            t[rows] *= outlier
            out[k] = t
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
            idx = torch.randperm(shape[0], generator=g)[:min(n_outlier, shape[0])]
            w[idx] *= outlier
            out[k] = w
        else:
            out[k] = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return out


def seeded_input(name: str, shape, seed: int, kind: str = "normal") -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 7919 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    if kind == "normal":
        return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if kind == "uniform":
        return torch.rand(tuple(shape), generator=g, dtype=torch.float32)
    raise ValueError(kind)
