"""SEVIR skill scores (CSI / POD / SUCR / BIAS) on the decoded frames -- reference ``SEVIRSkillScore``
(datasets/sevir/evaluation.py:88-285) without the torchmetrics dependency.

Same constructor keywords, ``update(pred, target)`` / ``compute()`` / ``reset()`` and result dictionary; the counting
(the only per-pixel work) is ONE launch of pd_sevir_skill_counts instead of the reference's per-threshold loop.  Counts
are exact int64, so scores match the reference bit for bit given the same frames.  `sync()` sums the counters over the
ranks of a process group (what torchmetrics' dist_reduce_fx="sum" does at compute time).
"""
import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L


class SEVIRSkillScore:
    def __init__(self, layout: str = "NHWT", mode: str = "0", seq_len: Optional[int] = None, preprocess_type: str = "sevir",
                 threshold_list: Sequence[int] = (16, 74, 133, 160, 181, 219),
                 metrics_list: Sequence[str] = ("csi", "bias", "sucr", "pod"), eps: float = 1e-4):
        if preprocess_type != "sevir":
            raise NotImplementedError("only preprocess_type='sevir' (no pooling) is implemented")
        if mode not in ("0", "1", "2"):
            raise NotImplementedError(f"mode {mode} not supported!")
        self.layout, self.mode, self.seq_len = layout, mode, seq_len
        self.threshold_list, self.metrics_list, self.eps = tuple(threshold_list), tuple(metrics_list), eps
        self.keep_seq_len_dim = mode in ("1", "2")
        if self.keep_seq_len_dim:
            assert isinstance(seq_len, int), "seq_len must be provided when we need to keep seq_len dim."
        self._counts = None          # int64 (n_thr, T or 1, 3) on the device of the first update
        self._thr = None

    # reference state attributes (float tensors there; exact integers here)
    @property
    def hits(self):
        return self._state(0)

    @property
    def misses(self):
        return self._state(1)

    @property
    def fas(self):
        return self._state(2)

    def _state(self, k):
        if self._counts is None:
            shape = (len(self.threshold_list), self.seq_len) if self.keep_seq_len_dim else (len(self.threshold_list),)
            return torch.zeros(shape)
        c = self._counts[..., k].float()
        return c if self.keep_seq_len_dim else c[:, 0]

    def reset(self):
        self._counts = None

    def update(self, pred: torch.Tensor, target: torch.Tensor):
        assert pred.shape == target.shape and pred.dim() == len(self.layout)
        if not pred.is_cuda:
            raise L.PrediffHipError("SEVIRSkillScore.update runs on the HIP device the frames were decoded on")
        ta = self.layout.find("T")
        T = pred.shape[ta]
        if self.keep_seq_len_dim:
            assert T == self.seq_len
        outer = int(np.prod(pred.shape[:ta])) if ta > 0 else 1
        inner = int(np.prod(pred.shape[ta + 1:])) if ta + 1 < pred.dim() else 1
        dev = pred.device
        if self._counts is None:
            self._counts = torch.zeros((len(self.threshold_list), T if self.keep_seq_len_dim else 1, 3), dtype=torch.int64, device=dev)
            self._thr = torch.tensor(self.threshold_list, dtype=torch.float32, device=dev)
        divisor = float(np.float32(1.0 / 255.0))       # data.float() / PREPROCESS_SCALE_01['vil'] (sevir_dataloader.py:679)
        with L.on_device(pred):
            L.sevir_skill_counts(pred.detach().float().contiguous(), target.detach().float().contiguous(), self._thr, divisor,
                                 self._counts, outer, T, inner, self.keep_seq_len_dim)

    def sync(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and self._counts is not None:
            dist.all_reduce(self._counts, op=dist.ReduceOp.SUM, group=group)

    @staticmethod
    def pod(hits, misses, fas, eps):
        return hits / (hits + misses + eps)

    @staticmethod
    def sucr(hits, misses, fas, eps):
        return hits / (hits + fas + eps)

    @staticmethod
    def csi(hits, misses, fas, eps):
        return hits / (hits + misses + fas + eps)

    @staticmethod
    def bias(hits, misses, fas, eps):
        b = (hits + fas) / (hits + misses + eps)
        return torch.pow(b / torch.log(torch.tensor(2.0)), 2.0)

    def compute(self):
        fn = {"pod": self.pod, "csi": self.csi, "sucr": self.sucr, "bias": self.bias}
        hits, misses, fas = self.hits.cpu(), self.misses.cpu(), self.fas.cpu()
        ret = {thr: {} for thr in self.threshold_list}
        ret["avg"] = {}
        for met in self.metrics_list:
            scores = fn[met](hits, misses, fas, self.eps).numpy()
            avg = np.zeros((self.seq_len,)) if self.keep_seq_len_dim else 0
            for i, thr in enumerate(self.threshold_list):
                score = scores[i] if self.keep_seq_len_dim else scores[i].item()
                ret[thr][met] = np.mean(score).item() if self.mode == "2" else score
                avg = avg + score
            avg = avg / len(self.threshold_list)
            ret["avg"][met] = np.mean(avg).item() if self.mode == "2" else avg
        return ret
