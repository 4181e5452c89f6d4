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


def seeded_input(name: str, shape, seed: int, kind: str = "normal") -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 7919 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    if kind == "normal":
        return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if kind == "uniform":
        return torch.rand(tuple(shape), generator=g, dtype=torch.float32)
    raise ValueError(kind)
