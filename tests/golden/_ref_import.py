"""Import the read-only reference (/root/reference) in THIS container only.

Test infrastructure for generating golden vectors (tests/golden/gen_golden.py).
The reference needs `lightning`, `diffusers` and `torchvision` at import time only
for base classes / isinstance checks / an LPIPS loss that the sampling path never
touches; they are absent here, so minimal stand-in *modules* are
registered in sys.modules before the import.  Nothing from the reference is copied.
This file never runs on the GPU box (no /root/reference there).
"""
import sys
import types

import torch
from torch import nn

REF_SRC = "/root/reference/src"


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install_stubs():
    if "lightning" in sys.modules and getattr(sys.modules["lightning"], "_prediff_stub", False):
        return
    lightning = _mod("lightning")
    lightning._prediff_stub = True

    class LightningDataModule:  # noqa: D401 - stand-in
        pass

    def seed_everything(seed, workers=False):
        torch.manual_seed(seed)
        return seed

    lightning.LightningDataModule = LightningDataModule
    lightning.seed_everything = seed_everything

    lp = _mod("lightning.pytorch")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    lp.LightningModule = LightningModule
    lp.LightningDataModule = LightningDataModule
    lp.seed_everything = seed_everything
    lightning.pytorch = lp
    lpu = _mod("lightning.pytorch.utilities")
    lpur = _mod("lightning.pytorch.utilities.rank_zero")
    lpur.rank_zero_only = lambda f: f
    lpu.rank_zero = lpur
    lp.utilities = lpu

    diffusers = _mod("diffusers")
    dm = _mod("diffusers.models")
    dma = _mod("diffusers.models.autoencoder_kl")

    class AutoencoderKLOutput:
        pass

    class DecoderOutput:
        pass

    dma.AutoencoderKLOutput = AutoencoderKLOutput
    dma.DecoderOutput = DecoderOutput
    dm.autoencoder_kl = dma
    diffusers.models = dm

    # torchmetrics.Metric / h5py: only needed to import datasets/sevir/evaluation.py (SEVIRSkillScore)
    tm = _mod("torchmetrics")

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default.clone()
            setattr(self, name, default.clone())

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

    tm.Metric = Metric
    _mod("h5py")

    tv = _mod("torchvision")
    tvm = _mod("torchvision.models")

    class VGG16_Weights:
        IMAGENET1K_V1 = None

    def vgg16(*a, **k):
        raise RuntimeError("torchvision stub: vgg16 is not available")

    tvm.VGG16_Weights = VGG16_Weights
    tvm.vgg16 = vgg16
    tv.models = tvm


def import_reference():
    """Returns a namespace with the reference classes used for golden generation."""
    install_stubs()
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    ns = types.SimpleNamespace()
    from prediff.models.cuboid_transformer import cuboid_transformer as ct
    from prediff.models.cuboid_transformer.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff.models import time_embed, utils as mutils
    from prediff.taming.autoencoder_kl import AutoencoderKL
    from prediff.taming import resnet as tresnet, attention as tattn
    from prediff.diffusion.latent_diffusion import LatentDiffusion
    from prediff.diffusion import utils as dutils
    from prediff.utils.distributions import DiagonalGaussianDistribution
    from prediff.diffusion.knowledge_alignment.sevir import SEVIRAvgIntensityAlignment
    ns.ct = ct
    ns.CuboidTransformerUNet = CuboidTransformerUNet
    ns.time_embed = time_embed
    ns.mutils = mutils
    ns.AutoencoderKL = AutoencoderKL
    ns.tresnet = tresnet
    ns.tattn = tattn
    ns.LatentDiffusion = LatentDiffusion
    ns.dutils = dutils
    ns.DiagonalGaussianDistribution = DiagonalGaussianDistribution
    ns.SEVIRAvgIntensityAlignment = SEVIRAvgIntensityAlignment
    from prediff.datasets.sevir.evaluation import SEVIRSkillScore
    ns.SEVIRSkillScore = SEVIRSkillScore
    return ns
