"""The reference's SEVIR script subclasses LatentDiffusion (scripts/prediff/sevirlr/train_sevirlr_prediff.py:70
``class PreDiffSEVIRPLModule(LatentDiffusion)``), calls ``super().__init__`` with the keyword set of :138-170, uses
``self.save_hyperparameters`` / ``self.log`` / ``self.local_rank`` / ``self.batch_axis`` and, in ``test_step`` (:905-979), ``self.get_input``
and ``self.sample``.  A module of that shape must work on prediff_amd.LatentDiffusion with only the imports swapped."""
import pytest
import torch
from torch import nn

from _cases import TINY_ALIGN_ARGS, TINY_UNET_CFGS, TINY_VAE_CFG
from _weights import seeded_input, seeded_state_dict
from prediff_amd import AutoencoderKL, CuboidTransformerUNet, LatentDiffusion
from prediff_amd.alignment import SEVIRAvgIntensityAlignment, get_alignment_kwargs_avg_x


class ScriptShapedModule(LatentDiffusion):
    """Same structure as PreDiffSEVIRPLModule: build the networks, hand them to LatentDiffusion.__init__, install the alignment."""

    def __init__(self, total_num_steps: int, precision="bf16"):
        cfg = TINY_UNET_CFGS["axial"]
        latent_model = CuboidTransformerUNet(**cfg, precision=precision)
        first_stage_model = AutoencoderKL(**TINY_VAE_CFG, precision=precision)
        super(ScriptShapedModule, self).__init__(
            torch_nn_module=latent_model, layout="NTHWC", data_shape=(2, 32, 32, 1), timesteps=1000, beta_schedule="linear",
            loss_type="l2", monitor="valid_loss_epoch", use_ema=False, log_every_t=100, clip_denoised=False, linear_start=1e-4,
            linear_end=2e-2, cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1.,
            parameterization="eps", learn_logvar=False, logvar_init=0., latent_shape=tuple(cfg["target_shape"]),
            first_stage_model=first_stage_model, cond_stage_model="__is_first_stage__", num_timesteps_cond=None,
            cond_stage_trainable=False, cond_stage_forward=None, scale_by_std=False, scale_factor=1.0)
        self.save_hyperparameters()
        self.alignment_obj = SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid",
                                                        model_args=dict(TINY_ALIGN_ARGS))
        self.alignment_model = self.alignment_obj.model
        self.set_alignment(alignment_fn=self.alignment_obj.get_mean_shift)
        self.total_num_steps = total_num_steps
        self.in_slice, self.out_slice = (slice(None), slice(0, 3)), (slice(None), slice(3, 5))
        self.logged = []

    def get_input(self, batch, **kwargs):
        seq = batch
        in_seq, out_seq = seq[self.in_slice], seq[self.out_slice].contiguous()
        return (out_seq, {"y": in_seq}, in_seq) if kwargs.get("return_verbose", False) else (out_seq, {"y": in_seq})

    def log(self, name, value, **kw):           # Lightning's logger needs a Trainer; the shim's log is a no-op either way
        self.logged.append(name)

    def test_step(self, batch, batch_idx):
        micro_batch_size = batch.shape[self.batch_axis]
        target_seq, cond, context_seq = self.get_input(batch, return_verbose=True)
        alignment_kwargs = get_alignment_kwargs_avg_x(context_seq=context_seq, target_seq=target_seq)
        aligned = self.sample(cond=cond, batch_size=micro_batch_size, return_intermediates=False, use_alignment=True,
                              alignment_kwargs=alignment_kwargs, verbose=False, timesteps=2).contiguous()
        pred = self.sample(cond=cond, batch_size=micro_batch_size, return_intermediates=False, verbose=False, timesteps=2).contiguous()
        self.log("test_mse_epoch", torch.mean((pred - target_seq) ** 2), prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)
        return aligned, pred, f"batch{batch_idx}_rank{self.local_rank}_sample0.npy"


def test_subclass_constructs_like_the_script_module():
    m = ScriptShapedModule(total_num_steps=10)
    assert isinstance(m, nn.Module) and isinstance(m, LatentDiffusion)
    try:
        from lightning.pytorch import LightningModule
        assert isinstance(m, LightningModule)            # with lightning installed the engine IS a LightningModule
    except ImportError:
        pass
    keys = list(m.state_dict().keys())
    assert any(k.startswith("torch_nn_module.first_proj.") for k in keys) and any(k.startswith("first_stage_model.encoder.") for k in keys)
    assert any(k.startswith("alignment_model.") for k in keys)
    assert m.device == torch.device("cpu") and m.local_rank == 0 and m.batch_axis == 0
    assert all(not p.requires_grad for p in m.first_stage_model.parameters())
    seq = torch.rand(2, 5, 32, 32, 1)
    tgt, cond, ctx = m.get_input(seq, return_verbose=True)
    assert tgt.shape == (2, 2, 32, 32, 1) and cond["y"].shape == (2, 3, 32, 32, 1)
    with pytest.raises(NotImplementedError):
        LatentDiffusion.get_input(m, seq)               # dataset dependent in the base class, as in the reference
    with pytest.raises(Exception):
        m.sample(cond={"y": ctx}, batch_size=2, timesteps=1)     # CPU tensors: the engine has no CPU path and says so


@pytest.mark.gpu
def test_subclass_test_step_on_gpu():
    m = ScriptShapedModule(total_num_steps=10)
    for mod, seed in ((m.torch_nn_module, 600), (m.first_stage_model, 601), (m.alignment_model, 700)):
        mod.load_state_dict(seeded_state_dict(mod.state_dict(), seed))
    m = m.cuda().eval()
    seq = seeded_input("dropin", (2, 5, 32, 32, 1), 3, kind="uniform").cuda()
    aligned, pred, name = m.test_step(seq, 0)
    assert aligned.shape == pred.shape == (2, 2, 32, 32, 1) and name == "batch0_rank0_sample0.npy"
    assert bool(torch.isfinite(aligned).all()) and bool(torch.isfinite(pred).all())
    assert m.logged == ["test_mse_epoch"] and m.device.type == "cuda"
