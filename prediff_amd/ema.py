"""Exponential moving average of the denoiser weights (reference utils/ema.py:6-77, `LitEma`): the shadow copy a PreDiff checkpoint
carries under `model_ema.*` and swaps in for validation / sampling (`LatentDiffusion.ema_scope`, latent_diffusion.py:280-293).

Same constructor, buffer naming and update rule as the reference so that its checkpoints load: one buffer per tracked parameter, named
like the parameter with the dots removed, plus `decay` (fp32 scalar) and `num_updates` (int32 scalar, -1 = constant decay).  One
difference that the sampling engine needs: prediff_amd freezes every parameter (no backward through the HIP kernels), so a module whose
parameters all have requires_grad = False is tracked in full instead of not at all (`track_frozen`, default: automatic).
"""
from typing import Dict, Iterable, List, Optional

import torch
from torch import nn


class LitEma(nn.Module):
    def __init__(self, model: nn.Module, decay: float = 0.9999, use_num_upates: bool = True, track_frozen: Optional[bool] = None):
        super().__init__()
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        params = list(model.named_parameters())
        if track_frozen is None:                      # the reference tracks p.requires_grad only; a fully frozen engine module is tracked whole
            track_frozen = not any(p.requires_grad for _, p in params)
        self.track_frozen = track_frozen
        self.m_name2s_name: Dict[str, str] = {}
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0 if use_num_upates else -1, dtype=torch.int))
        for name, p in params:
            if self._tracked(p):
                shadow = name.replace(".", "")        # '.' is not allowed in a buffer name
                self.m_name2s_name[name] = shadow
                self.register_buffer(shadow, p.detach().clone())
        self.collected_params: List[torch.Tensor] = []

    def _tracked(self, p: torch.Tensor) -> bool:
        return p.requires_grad or self.track_frozen

    @torch.no_grad()
    def forward(self, model: nn.Module):
        """shadow -= (1 - decay_t) * (shadow - param) with decay_t = min(decay, (1 + n) / (10 + n)) after n updates (utils/ema.py:25-44)."""
        decay = self.decay
        if self.num_updates >= 0:
            self.num_updates += 1
            decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus_decay = 1.0 - decay
        shadows = dict(self.named_buffers())
        for name, p in model.named_parameters():
            if self._tracked(p):
                s = shadows[self.m_name2s_name[name]]
                s.sub_(one_minus_decay * (s - p.to(s.dtype)))
            else:
                assert name not in self.m_name2s_name

    @torch.no_grad()
    def copy_to(self, model: nn.Module):
        shadows = dict(self.named_buffers())
        for name, p in model.named_parameters():
            if self._tracked(p):
                # p.copy_, not p.data.copy_ (the reference's form): a write through .data leaves p._version alone, and the HIP engine keys
                # its packed operands on (data_ptr, _version) -- it would go on computing with the previous weights
                p.copy_(shadows[self.m_name2s_name[name]])
            else:
                assert name not in self.m_name2s_name

    def store(self, parameters: Iterable[torch.Tensor]):
        """Keep a copy of `parameters` (to `restore` after validating / sampling with the EMA weights)."""
        self.collected_params = [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters: Iterable[torch.Tensor]):
        for saved, p in zip(self.collected_params, parameters):
            p.copy_(saved)                # (bumps p._version: see copy_to)
