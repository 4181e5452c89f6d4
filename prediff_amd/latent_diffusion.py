"""Latent-diffusion sampling driver on MI355X: the reference's ``LatentDiffusion`` sampling API
(reference diffusion/latent_diffusion.py: register_schedule :228-278, sample :686-724, p_sample_loop :633-684,
p_sample :598-631, p_mean_variance :568-590, q_posterior :559-566, predict_start_from_noise :553-557,
aligned_mean :592-596, apply_model :440-445, encode/decode_first_stage :423-438, cond-stage wrapper :361-380).

Same constructor keywords and call conventions; what changes:
  * one denoising step = denoiser forward (HIP kernels) + ONE fused step-epilogue kernel (pd_ddpm_step / pd_ddim_step)
    instead of ~10 elementwise launches; the per-step schedule scalars are gathered on the device from t;
  * the step (denoiser + epilogue) is captured once into a HIP graph per (batch, sampler) and replayed: the host loop
    only refreshes t / noise, so small batches are not launch-bound;
  * a DDIM sampler exists (the reference ships only the two schedule helpers): ``sample(...,
    sampler="ddim", ddim_steps=50, eta=0.)``;
  * of the training side, everything that needs no gradient is here (q_sample, forward / p_losses as loss VALUES, LitEma + ema_scope,
    validation_step); training_step raises: there is no backward pass through the HIP kernels.
The knowledge-alignment hook (`set_alignment`) is honoured with the reference semantics; its gradient stays in
PyTorch autograd (north_star) and therefore runs outside the captured graph.
"""
from typing import Any, Callable, Dict, Optional, Sequence, Union

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .distributions import DiagonalGaussianDistribution
from .ema import LitEma
from .loss_eval import LossEvaluationMixin
from .schedule import make_beta_schedule, make_ddim_sampling_parameters, make_ddim_timesteps, schedule_tables


def parse_layout_shape(layout: str) -> Dict[str, int]:
    """utils/layout.py:18-41"""
    return {"batch_axis": layout.find("N"), "t_axis": layout.find("T"), "h_axis": layout.find("H"),
            "w_axis": layout.find("W"), "c_axis": layout.find("C")}


def _disabled_train(self, mode=True):
    return self


def _module_base():
    """The reference's LatentDiffusion is a ``pl.LightningModule`` (diffusion/latent_diffusion.py:25) and the SEVIR script subclasses
    it (scripts/prediff/sevirlr/train_sevirlr_prediff.py:70 ``class PreDiffSEVIRPLModule(LatentDiffusion)``) and hands the object to
    ``Trainer.test``.  With lightning installed the engine derives from the real LightningModule, so that subclass works unchanged;
    without it (this image) a plain nn.Module with the few LightningModule members the script's sampling path touches
    (save_hyperparameters / log / log_dict no-ops, device, local_rank, global_rank, current_epoch)."""
    try:
        from lightning.pytorch import LightningModule      # noqa: WPS433
        if isinstance(LightningModule, type) and issubclass(LightningModule, nn.Module):
            return LightningModule
    except Exception:
        pass

    class _LightningModuleShim(nn.Module):
        local_rank = 0
        global_rank = 0
        current_epoch = 0
        global_step = 0

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        @property
        def device(self):
            for t in list(self.parameters(recurse=True))[:1] + list(self.buffers(recurse=True))[:1]:
                return t.device
            return torch.device("cpu")

    return _LightningModuleShim


def _on_own_device(fn):
    """Run a sampling entry point with the module's device current (kernels are launched on the current stream of the current
    device; HIP graphs and lane streams are created with an explicit device, and both must agree)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = self.betas.device
        if dev.type != "cuda":
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapped


class LatentDiffusion(LossEvaluationMixin, _module_base()):

    def __init__(self, torch_nn_module: nn.Module, layout: str = "NTHWC", data_shape: Sequence[int] = (10, 128, 128, 4),
                 timesteps=1000, beta_schedule="linear", loss_type="l2", monitor="val/loss", use_ema=True,
                 log_every_t=100, clip_denoised=False, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3,
                 given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1., parameterization="eps",
                 learn_logvar=False, logvar_init=0.,
                 # latent diffusion
                 latent_shape: Sequence[int] = (10, 16, 16, 4), first_stage_model: nn.Module = None,
                 cond_stage_model: Union[str, nn.Module] = None, num_timesteps_cond=None, cond_stage_trainable=False,
                 cond_stage_forward=None, scale_by_std=False, scale_factor=1.0):
        super().__init__()
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        if layout != "NTHWC":
            raise NotImplementedError("the HIP engine works on channels-last NTHWC latents")
        self.parameterization = parameterization
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.torch_nn_module = torch_nn_module
        self.layout = layout
        self.data_shape = tuple(data_shape)
        p = parse_layout_shape(layout)
        self.batch_axis, self.t_axis, self.h_axis, self.w_axis, self.c_axis = (p[k] for k in ("batch_axis", "t_axis", "h_axis", "w_axis", "c_axis"))
        self.use_ema = use_ema          # EMA shadow weights are a training feature; sampling uses the raw weights
        self.v_posterior = v_posterior
        self.original_elbo_weight, self.l_simple_weight, self.loss_type, self.monitor = original_elbo_weight, l_simple_weight, loss_type, monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.learn_logvar = learn_logvar
        self.register_buffer("logvar", torch.full(fill_value=float(logvar_init), size=(self.num_timesteps,)))
        self.loss_mean_dim = tuple(a for a in range(len(self.data_shape) + 1) if a != self.batch_axis)     # every axis but the batch axis
        if self.use_ema:
            self.model_ema = LitEma(self.torch_nn_module)          # shadow weights: `model_ema.*` of a reference checkpoint
        self.latent_shape = tuple(latent_shape)
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        assert self.num_timesteps_cond <= timesteps
        # num_timesteps_cond > 1: the conditioning latents are re-noised in front of every ancestral step (latent_diffusion.py:155-157,
        # 295-299, 665-667; dead at every shipped config, "TODO: drop this option" in the reference) -- eager DDPM loop only
        self.shorten_cond_schedule = self.num_timesteps_cond > 1
        if self.shorten_cond_schedule:
            self.make_cond_schedule()
        self.cond_stage_trainable = False
        self.scale_by_std = scale_by_std
        if not scale_by_std:
            self.scale_factor = float(scale_factor)
        else:
            # a persistent buffer, as in the reference (latent_diffusion.py:160-164): a checkpoint written with scale_by_std=True carries
            # "scale_factor" and has to load with strict=True.  (The std-rescaling of the very first training batch, :301-317, is training-side.)
            self.register_buffer("scale_factor", torch.tensor(float(scale_factor)))
        self.alignment_fn: Optional[Callable] = None
        self.instantiate_first_stage(first_stage_model)
        self.instantiate_cond_stage(cond_stage_model, cond_stage_forward)
        self._graphs: Dict = {}
        self.use_hip_graph = True
        # lanes: the batch is advanced as `num_streams` equal sub-batches, each a HIP graph replayed on its own stream with its own
        # workspace.  Trajectories are independent, so the sub-batches fill each other's idle CUs (tile-count quantisation, HBM-bound
        # phases of the fused kernels): +12...20 % throughput at 32-64 trajectories.  precision="fp32" (never splits K): results
        # are bit-identical to num_streams = 1.  bf16 / fp8 with <= 16 trajectories per launch: the split-K Conv3d (csrc/igemm256.hip)
        # picks its K slicing from the per-launch batch, so another lane count / micro-batch / ensemble sharding changes the fp32
        # summation order and the latents agree to bf16 noise (5e-3 per forward), not bit for bit (DESIGN.md §4;
        # tests/test_hip_configs.py::test_v1_lane_split_tolerance).  `torch_nn_module.split_k = False` is the reproducible mode.
        # Default: two lanes from 8 trajectories per lane on, ONE below that (MI355X, round 5: 16 trajectories 1345-1358 steps/s in two lanes vs
        # 1293-1325 in one, but 12: 1020 vs 1089 and 8: 918 vs 951 -- profiles/r05_r_small_batch_lanes.txt).  Assigning `num_streams` pins it.
        self._num_streams, self._lanes_pinned = 2, False
        self.aligned_lanes = 1        # knowledge-aligned loop: denoiser lanes next to the guidance stream (see p_sample_loop)
        self.guidance_high_priority = False   # knowledge-aligned loop: run the guidance on a high-priority side stream (A/B switch;
                                              # measured neutral: 32.9 vs 32.8-33.1 ms at 32 trajectories, 11.5 vs 11.6 ms at 8)
        self._guidance_streams: Dict = {}
        self._lane_streams: Dict = {}

    # ------------------------------------------------------------------------------------------------ schedule
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = np.asarray(given_betas, dtype=np.float64) if given_betas is not None else \
            make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        tabs = schedule_tables(betas, self.v_posterior)
        for k, v in tabs.items():
            self.register_buffer(k, torch.tensor(v, dtype=torch.float32))
        # device table consumed by pd_ddpm_step: [sqrt_recip | sqrt_recipm1 | coef1 | coef2 | logvar_clipped]
        self.register_buffer("_step_coef", torch.stack([
            self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
            self.posterior_mean_coef2, self.posterior_log_variance_clipped]).contiguous(), persistent=False)
        self._alphas_cumprod_f64 = np.cumprod(1.0 - betas)
        # weights of the variational-bound term of the training / validation loss (latent_diffusion.py:270-278), in the reference's own
        # fp32 tensor arithmetic
        alphas32 = torch.tensor(1.0 - betas, dtype=torch.float32)
        if self.parameterization == "eps":
            lvlb = self.betas ** 2 / (2 * self.posterior_variance * alphas32 * (1 - self.alphas_cumprod))
        else:
            ac32 = torch.tensor(np.cumprod(1.0 - betas), dtype=torch.float32)
            lvlb = 0.5 * torch.sqrt(ac32) / (2.0 * 1 - ac32)
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", lvlb, persistent=False)

    def set_alignment(self, alignment_fn: Callable = None):
        """alignment_fn(zt, t, zc=None, y=None, **kwargs) -> tensor like zt (latent_diffusion.py:169-180)."""
        self.alignment_fn = alignment_fn

    def extract_into_tensor(self, a, t, x_shape):
        out = a.gather(-1, t)
        shape = [1] * len(x_shape)
        shape[self.batch_axis] = t.shape[0]
        return out.reshape(shape)

    def get_batch_latent_shape(self, batch_size=1):
        s = list(self.latent_shape)
        s.insert(self.batch_axis, batch_size)
        return tuple(s)

    def get_batch_data_shape(self, batch_size=1):
        s = list(self.data_shape)
        s.insert(self.batch_axis, batch_size)
        return tuple(s)

    # ------------------------------------------------------------------------------------------------ first / cond stage
    def instantiate_first_stage(self, first_stage_model):
        if first_stage_model is None:
            self.first_stage_model = None
            return
        assert isinstance(first_stage_model, nn.Module)
        self.first_stage_model = first_stage_model.eval()
        self.first_stage_model.train = _disabled_train.__get__(self.first_stage_model)
        for p in self.first_stage_model.parameters():
            p.requires_grad = False

    def instantiate_cond_stage(self, cond_stage_model, cond_stage_forward):
        if cond_stage_model is None:
            self.cond_stage_model, self.cond_stage_forward = None, None
            return
        is_first = isinstance(cond_stage_model, str) and cond_stage_model == "__is_first_stage__"
        if is_first:
            model = self.first_stage_model
        elif isinstance(cond_stage_model, nn.Module):
            model = cond_stage_model
        else:
            raise NotImplementedError
        self.cond_stage_model = model
        for p in model.parameters():
            p.requires_grad = False
        if cond_stage_forward is None:
            fwd = model.encode if hasattr(model, "encode") and callable(model.encode) else model.__call__
        else:
            fwd = getattr(model, cond_stage_forward)

        def func(c):
            if is_first:
                c = c.get("y")
                b = c.shape[self.batch_axis]
                c = self._to_frames(c)
            c = fwd(c)
            if isinstance(c, DiagonalGaussianDistribution):
                c = c.mode()
            elif hasattr(c, "latent_dist"):
                c = c.latent_dist.mode()
            if is_first:
                c = self._from_frames(c, b)
            return c
        self.cond_stage_forward = func

    @staticmethod
    def _to_frames(x):      # "N T H W C -> (N T) C H W"
        N, T, H, W, C = x.shape
        return x.permute(0, 1, 4, 2, 3).reshape(N * T, C, H, W)

    @staticmethod
    def _from_frames(x, N):  # "(N T) C H W -> N T H W C"
        NT, C, H, W = x.shape
        return x.reshape(N, NT // N, C, H, W).permute(0, 1, 3, 4, 2).contiguous()

    @property
    def einops_layout(self):
        return " ".join(self.layout)

    @property
    def einops_spatial_layout(self):
        return "(N T) C H W"

    @torch.no_grad()
    def decode_first_stage(self, z):
        z = 1.0 / self.scale_factor * z
        b = z.shape[self.batch_axis]
        out = self.first_stage_model.decode(self._to_frames(z))
        if hasattr(out, "sample") and not isinstance(out, torch.Tensor):
            out = out.sample
        return self._from_frames(out, b)

    @torch.no_grad()
    def encode_first_stage(self, x):
        post = self.first_stage_model.encode(x)
        if isinstance(post, DiagonalGaussianDistribution):
            z = post.sample()
        elif isinstance(post, torch.Tensor):
            z = post
        else:
            z = post.latent_dist.sample()
        return (self.scale_factor * z).detach()

    def get_first_stage_encoding(self, encoder_posterior):
        """latent_diffusion.py:382-391"""
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        elif hasattr(encoder_posterior, "latent_dist"):
            z = encoder_posterior.latent_dist.sample()
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    def get_input(self, batch, **kwargs):
        """Dataset dependent (latent_diffusion.py:405-421): subclasses return (target, {"y": context}[, context])."""
        raise NotImplementedError("get_input is dataset dependent: re-implement it in the subclass "
                                  "(e.g. train_sevirlr_prediff.py:733-759)")

    # loss evaluation / EMA / validation_step: loss_eval.LossEvaluationMixin

    def apply_model(self, x_noisy, t, cond):
        out = self.torch_nn_module(x_noisy, t, cond)
        return out[0] if isinstance(out, tuple) else out

    # ------------------------------------------------------------------------------------------------ reference-API math (torch ops)
    def make_cond_schedule(self):
        """cond_ids[i] = the noise level the conditioning latents get in front of step i: num_timesteps_cond levels spread evenly
        over the schedule for the first num_timesteps_cond steps, the last level for every later one (latent_diffusion.py:295-299)."""
        last = self.num_timesteps - 1
        levels = torch.linspace(0, last, self.num_timesteps_cond).round().long()
        ids = torch.full((self.num_timesteps,), last, dtype=torch.long)
        ids[:self.num_timesteps_cond] = levels
        self.register_buffer("cond_ids", ids)

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (self.extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                self.extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def predict_start_from_noise(self, x_t, t, noise):
        return (self.extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t -
                self.extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (self.extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start +
                self.extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, self.extract_into_tensor(self.posterior_variance, t, x_t.shape),
                self.extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape))

    def p_mean_variance(self, zt, zc, t, clip_denoised: bool, return_x0=False, score_corrector=None, corrector_kwargs=None):
        model_out = self.apply_model(zt, t, zc)
        if score_corrector is not None:
            assert self.parameterization == "eps"
            model_out = score_corrector.modify_score(self, model_out, zt, t, zc, **corrector_kwargs)
        z_recon = self.predict_start_from_noise(zt, t=t, noise=model_out) if self.parameterization == "eps" else model_out
        if clip_denoised:
            z_recon = z_recon.clamp(-1.0, 1.0)
        mean, var, logvar = self.q_posterior(x_start=z_recon, x_t=zt, t=t)
        return (mean, var, logvar, z_recon) if return_x0 else (mean, var, logvar)

    def aligned_mean(self, zt, t, zc, y, orig_mean, orig_log_var, **kwargs):
        return orig_mean - (0.5 * orig_log_var).exp() * self.alignment_fn(zt, t, zc=zc, y=y, **kwargs)

    # ------------------------------------------------------------------------------------------------ one step
    @torch.no_grad()
    @_on_own_device
    def p_sample(self, zt, zc, t, y=None, use_alignment=False, alignment_kwargs=None, clip_denoised=False, return_x0=False,
                 temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None, noise=None):
        """One ancestral step.  `noise=None` draws torch.randn on zt's device exactly where the reference does (:620)."""
        if return_x0 or noise_dropout > 0. or score_corrector is not None or self.parameterization != "eps":
            return self._p_sample_torch(zt, zc, t, y, use_alignment, alignment_kwargs, clip_denoised, return_x0, temperature,
                                        noise_dropout, score_corrector, corrector_kwargs, noise)
        eps = self.apply_model(zt, t, zc)
        shift = None
        if use_alignment:
            shift = self.alignment_fn(zt, t, zc=zc, y=y, **(alignment_kwargs or {})).contiguous().float()
        if noise is None:
            noise = torch.randn(zt.shape, device=zt.device)
        return self._ddpm_update(zt, eps, noise, shift, t, temperature, clip_denoised)

    def _ddpm_update(self, zt, eps, noise, shift, t, temperature=1.0, clip_denoised=False, out=None):
        B = zt.shape[0]
        out = torch.empty_like(zt) if out is None else out
        with L.on_device(zt):
            L.ddpm_step(zt.contiguous(), eps.contiguous(), noise.contiguous(), shift, t.to(torch.int64).contiguous(),
                            self._step_coef, self.num_timesteps, out, B, zt[0].numel(), temperature, clip_denoised)
        return out

    def _p_sample_torch(self, zt, zc, t, y, use_alignment, alignment_kwargs, clip_denoised, return_x0, temperature,
                        noise_dropout, score_corrector, corrector_kwargs, noise):
        outs = self.p_mean_variance(zt=zt, zc=zc, t=t, clip_denoised=clip_denoised, return_x0=return_x0,
                                    score_corrector=score_corrector, corrector_kwargs=corrector_kwargs)
        mean, _, logvar = outs[:3]
        if use_alignment:
            mean = self.aligned_mean(zt=zt, t=t, zc=zc, y=y, orig_mean=mean, orig_log_var=logvar, **(alignment_kwargs or {}))
        noise = (torch.randn(zt.shape, device=zt.device) if noise is None else noise) * temperature
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        shape = [1] * zt.dim()
        shape[self.batch_axis] = zt.shape[self.batch_axis]
        nonzero = (1 - (t == 0).float()).reshape(shape)
        res = mean + nonzero * (0.5 * logvar).exp() * noise
        return (res, outs[3]) if return_x0 else res

    # ------------------------------------------------------------------------------------------------ graph-captured step
    def _graph_step(self, kind, B, zc, device, lane=0):
        """Capture (denoiser forward + step epilogue) once per (lane, kind, B) into a HIP graph; returns the static buffers.
        `lane` selects the denoiser's workspace set, so graphs of different lanes may replay concurrently on different streams."""
        net = self.torch_nn_module
        if hasattr(net, "_ensure_packed"):
            net._ensure_packed(device)       # a weight update re-packs -> new operand buffers -> the old graph is stale
        key = (kind, B, str(device), tuple(zc.shape), getattr(net, "_pack_generation", None), self.clip_denoised)
        g = self._graphs.get(lane)
        if g is not None and g[0] == key:
            g[1]["zc"].copy_(zc)
            return g[1]
        shape = self.get_batch_latent_shape(B)
        st = dict(z=torch.zeros(shape, device=device), noise=torch.zeros(shape, device=device),
                  t=torch.zeros(B, dtype=torch.int64, device=device), out=torch.zeros(shape, device=device),
                  coef=torch.zeros(B, 3, device=device), zc=zc.detach().clone().float().contiguous())

        def body():
            if hasattr(net, "_ws_slot"):
                net._ws_slot = lane
            eps = self.apply_model(st["z"], st["t"], st["zc"])
            if kind == "eps":            # denoiser only: the step epilogue needs the alignment shift, computed outside the graph
                st["out"].copy_(eps)
            elif kind == "ddpm":
                self._ddpm_update(st["z"], eps, st["noise"], None, st["t"], 1.0, self.clip_denoised, out=st["out"])
            else:
                L.ddim_step(st["z"], eps, st["noise"], st["coef"], st["out"], B, st["z"][0].numel())
        torch.cuda.synchronize(device)          # no other lane is mid-replay while this one warms up / captures
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(2):          # warm-up: weight packing, workspace allocation, attribute calls
                body()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        if hasattr(net, "_ws_slot"):
            net._ws_slot = 0
        st["graph"] = graph
        self._graphs[lane] = (key, st)   # one live graph per lane
        return st

    MIN_AUTO_LANE_BATCH = 8

    @property
    def num_streams(self):
        return self._num_streams

    @num_streams.setter
    def num_streams(self, value):
        self._num_streams, self._lanes_pinned = int(value), True

    def _lanes(self, kind, B, cond, device, allow):
        """Per-lane graph states and streams for a batch of B, or None when the single-graph path has to be used."""
        S = int(self.num_streams) if allow else 1
        if not self._lanes_pinned and S > 1 and B // S < self.MIN_AUTO_LANE_BATCH:
            S = 1
        if S <= 1 or B % S or B // S < 1 or not isinstance(cond, torch.Tensor):
            return None
        Bl = B // S
        key = str(device)
        if len(self._lane_streams.get(key, ())) != S:
            self._lane_streams[key] = [torch.cuda.Stream(device=device) for _ in range(S)]
        sts = [self._graph_step(kind, Bl, cond[l * Bl:(l + 1) * Bl], device, lane=l) for l in range(S)]
        return sts, self._lane_streams[key], Bl

    def _guidance_stream(self, device):
        key = str(device)
        if key not in self._guidance_streams:
            self._guidance_streams[key] = torch.cuda.Stream(device=device, priority=-1)
        return self._guidance_streams[key]

    @staticmethod
    def _lane_step(sts, streams, Bl, device, fill, keep=(), advance=True):
        """One step of every lane: `fill(st, sl)` writes the lane's inputs (slice sl of the batch) on the lane's stream, then the
        lane's graph is replayed and its latent advanced in place.  `keep`: tensors made on the caller's stream that `fill` reads."""
        main = torch.cuda.current_stream(device)
        for l, (st, stream) in enumerate(zip(sts, streams)):
            stream.wait_stream(main)                  # inputs produced on the caller's stream (noise draws, schedules)
            with torch.cuda.stream(stream):
                fill(st, slice(l * Bl, (l + 1) * Bl))
                st["graph"].replay()
                if advance:
                    st["z"].copy_(st["out"])
            for t in keep:
                if t is not None:
                    t.record_stream(stream)           # the caching allocator must not recycle it before the lane has read it

    # ------------------------------------------------------------------------------------------------ loops
    @torch.no_grad()
    @_on_own_device
    def p_sample_loop(self, cond, shape, y=None, use_alignment=False, alignment_kwargs=None, return_intermediates=False,
                      x_T=None, verbose=False, callback=None, timesteps=None, mask=None, x0=None, img_callback=None,
                      start_T=None, log_every_t=None, noise_tape=None):
        """Ancestral loop t = timesteps-1 .. 0 (latent_diffusion.py:633-684).  RNG draw order as the reference:
        [x_T, noise_{T-1}, ..., noise_0] from the device's default generator, unless `noise_tape` supplies them.  With
        shorten_cond_schedule every step draws the conditioning noise first: [x_T, c_{T-1}, noise_{T-1}, ..., c_0, noise_0]."""
        log_every_t = log_every_t or self.log_every_t
        device = self.betas.device
        B = shape[self.batch_axis]
        if x_T is not None:
            img = x_T
            if noise_tape is not None:
                noise_tape[0]             # a step-ordered (lazy) tape still owes draw 0 = x_T: consume it, keep the caller's x_T
        elif noise_tape is not None:
            img = noise_tape[0].to(device)
        else:
            img = torch.randn(shape, device=device)
        intermediates = [img]
        timesteps = self.num_timesteps if timesteps is None else timesteps
        if start_T is not None:
            timesteps = min(timesteps, start_T)
        if mask is not None:
            assert x0 is not None
        shorten = self.shorten_cond_schedule          # the condition changes every step: eager path
        use_graph = (self.use_hip_graph and not use_alignment and self.parameterization == "eps" and img.is_cuda
                     and isinstance(cond, torch.Tensor) and not shorten)   # a dict / None condition cannot be a static graph input: eager path
        lanes = self._lanes("ddpm", B, cond, device, use_graph and mask is None and callback is None and img_callback is None
                            and not return_intermediates)
        if lanes is not None:
            # independent sub-batches, one HIP stream each; noise is drawn for the whole batch (reference RNG order) and sliced
            sts, streams, Bl = lanes
            for l, st in enumerate(sts):
                st["z"].copy_(img[l * Bl:(l + 1) * Bl])
            for k, i in enumerate(reversed(range(0, timesteps))):
                noise = noise_tape[1 + k].to(device) if noise_tape is not None else torch.randn(shape, device=device)

                def fill(st, sl, i=i, noise=noise):
                    st["t"].fill_(i)
                    st["noise"].copy_(noise[sl])
                self._lane_step(sts, streams, Bl, device, fill, keep=(noise,))
            for stream in streams:
                torch.cuda.current_stream(device).wait_stream(stream)
            return torch.cat([st["z"] for st in sts], dim=0)
        # knowledge alignment: the guidance gradient depends on z_t only, not on eps, so it runs (PyTorch autograd, caller's stream)
        # concurrently with the denoiser graphs of the lanes; the step epilogue joins them.  Same arithmetic as the eager path.
        eps_lanes = None
        if use_alignment and self.use_hip_graph and self.parameterization == "eps" and img.is_cuda and isinstance(cond, torch.Tensor) and not shorten:
            # one denoiser graph on one side stream: the guidance is the concurrent second stream of work, and splitting the denoiser
            # into lanes as well only makes the three compete (measured at 32 / 8 trajectories: 33.4 / 11.6 ms per step with one
            # lane, 34.8 / 13.6 with two, 37.2 / 16.0 with four -- profiles/r02_j_time_alignment*.log).  `aligned_lanes` overrides.
            saved = (self._num_streams, self._lanes_pinned)
            try:                                 # an exception in the capture must not leave the module with another lane count
                self.num_streams = max(1, int(self.aligned_lanes))
                if B % max(1, self.num_streams):
                    self.num_streams = 1
                eps_lanes = self._lanes("eps", B, cond, device, True) if self.num_streams > 1 else None
            finally:
                self._num_streams, self._lanes_pinned = saved
            if eps_lanes is None:
                key = str(device)
                if len(self._lane_streams.get(key, ())) < 1:
                    self._lane_streams[key] = [torch.cuda.Stream(device=device)]
                eps_lanes = ([self._graph_step("eps", B, cond, device, lane=0)], self._lane_streams[key][:1], B)
        st = self._graph_step("ddpm", B, cond, device) if use_graph else None
        for k, i in enumerate(reversed(range(0, timesteps))):
            if shorten:
                ts = torch.full((B,), i, device=device, dtype=torch.long)
                cn = noise_tape[1 + 2 * k].to(device) if noise_tape is not None else torch.randn_like(cond)
                cond = self.q_sample(cond, self.cond_ids[ts], noise=cn)       # (cumulative, as in the reference: `cond` is overwritten)
                noise = noise_tape[2 + 2 * k].to(device) if noise_tape is not None else None
            else:
                noise = noise_tape[1 + k].to(device) if noise_tape is not None else None
            if eps_lanes is not None:
                sts, streams, Bl = eps_lanes
                ts = torch.full((B,), i, device=device, dtype=torch.long)
                cur = img

                def fill(lst, sl, i=i, cur=cur):
                    lst["z"].copy_(cur[sl])
                    lst["t"].fill_(i)
                self._lane_step(sts, streams, Bl, device, fill, keep=(cur,), advance=False)
                if self.guidance_high_priority:
                    # the guidance on its own HIGH-priority stream: its many small kernels are dispatched ahead of the denoiser
                    # graph's queued workgroups whenever a CU has room, instead of taking turns with whole kernels
                    main = torch.cuda.current_stream(device)
                    gs = self._guidance_stream(device)
                    gs.wait_stream(main)
                    with torch.cuda.stream(gs):
                        shift = self.alignment_fn(cur, ts, zc=cond, y=y, **(alignment_kwargs or {})).contiguous().float()
                    for tns in (cur, ts):
                        tns.record_stream(gs)
                    main.wait_stream(gs)
                    shift.record_stream(main)
                else:
                    shift = self.alignment_fn(cur, ts, zc=cond, y=y, **(alignment_kwargs or {})).contiguous().float()
                for stream in streams:
                    torch.cuda.current_stream(device).wait_stream(stream)
                eps = sts[0]["out"] if len(sts) == 1 else torch.cat([lst["out"] for lst in sts], dim=0)
                if noise is None:
                    noise = torch.randn(shape, device=device)
                img = self._ddpm_update(cur, eps, noise, shift, ts, 1.0, self.clip_denoised)
            elif st is not None:
                st["z"].copy_(img)
                st["t"].fill_(i)
                st["noise"].copy_(noise) if noise is not None else st["noise"].normal_()
                st["graph"].replay()
                img = st["out"].clone()
            else:
                ts = torch.full((B,), i, device=device, dtype=torch.long)
                img = self.p_sample(zt=img, zc=cond, t=ts, y=y, use_alignment=use_alignment, alignment_kwargs=alignment_kwargs,
                                    clip_denoised=self.clip_denoised, noise=noise)
            if mask is not None:
                ts = torch.full((B,), i, device=device, dtype=torch.long)
                img = self.q_sample(x0, ts) * mask + (1.0 - mask) * img
            if i % log_every_t == 0 or i == timesteps - 1:
                intermediates.append(img)
            if callback:
                callback(i)
            if img_callback:
                img_callback(img, i)
        return (img, intermediates) if return_intermediates else img

    @torch.no_grad()
    @_on_own_device
    def ddim_sample_loop(self, cond, shape, ddim_steps=50, eta=0.0, x_T=None, noise_tape=None, return_intermediates=False,
                         ddim_discretize="uniform"):
        """DDIM over the reference's timestep subset (diffusion/utils.py:42-70).  NOT in the reference:
        z_prev = sqrt(a_prev) z0 + sqrt(1 - a_prev - sigma^2) eps + sigma n, denoiser queried at t = steps[i]."""
        device = self.betas.device
        B = shape[self.batch_axis]
        if self.shorten_cond_schedule:
            raise NotImplementedError("shorten_cond_schedule (num_timesteps_cond > 1) is defined for the ancestral loop only")
        if not (1 <= int(ddim_steps) <= self.num_timesteps):
            raise ValueError(f"ddim_steps must be in [1, {self.num_timesteps}], got {ddim_steps}")
        if self.clip_denoised:
            raise NotImplementedError("clip_denoised=True is defined for the ancestral sampler only (the DDIM step has no clamp)")
        steps = np.minimum(make_ddim_timesteps(ddim_discretize, ddim_steps, self.num_timesteps), self.num_timesteps - 1)
        sig, a, a_prev = make_ddim_sampling_parameters(self._alphas_cumprod_f64.astype(np.float32).astype(np.float64), steps, eta)
        if x_T is not None:
            img = x_T
            if noise_tape is not None:
                noise_tape[0]             # see p_sample_loop
        elif noise_tape is not None:
            img = noise_tape[0].to(device)
        else:
            img = torch.randn(shape, device=device)
        intermediates = [img]
        lanes = self._lanes("ddim", B, cond, device, self.use_hip_graph and img.is_cuda and not return_intermediates)
        if lanes is not None:
            sts, streams, Bl = lanes
            for l, st in enumerate(sts):
                st["z"].copy_(img[l * Bl:(l + 1) * Bl])
            for k, idx in enumerate(reversed(range(len(steps)))):
                coef = torch.tensor([[a[idx], a_prev[idx], sig[idx]]], dtype=torch.float32).repeat(Bl, 1).to(device)
                noise = None
                if noise_tape is not None:
                    noise = noise_tape[1 + k].to(device)
                elif eta > 0:
                    noise = torch.randn(shape, device=device)

                def fill(st, sl, t=int(steps[idx]), coef=coef, noise=noise):
                    st["t"].fill_(t)
                    st["coef"].copy_(coef)
                    st["noise"].copy_(noise[sl]) if noise is not None else st["noise"].zero_()
                self._lane_step(sts, streams, Bl, device, fill, keep=(coef, noise))
            for stream in streams:
                torch.cuda.current_stream(device).wait_stream(stream)
            return torch.cat([st["z"] for st in sts], dim=0)
        # a dict / None condition cannot be a static graph input: eager path
        st = self._graph_step("ddim", B, cond, device) if (self.use_hip_graph and img.is_cuda and isinstance(cond, torch.Tensor)) else None
        for k, idx in enumerate(reversed(range(len(steps)))):
            coef = torch.tensor([[a[idx], a_prev[idx], sig[idx]]], dtype=torch.float32).repeat(B, 1)
            noise = None
            if noise_tape is not None:
                noise = noise_tape[1 + k].to(device)
            elif eta > 0:
                noise = torch.randn(shape, device=device)
            if st is not None:
                st["z"].copy_(img)
                st["t"].fill_(int(steps[idx]))
                st["coef"].copy_(coef)
                st["noise"].copy_(noise) if noise is not None else st["noise"].zero_()
                st["graph"].replay()
                img = st["out"].clone()
            else:
                ts = torch.full((B,), int(steps[idx]), device=device, dtype=torch.long)
                eps = self.apply_model(img, ts, cond)
                out = torch.empty_like(img)
                L.ddim_step(img.contiguous(), eps, noise if noise is not None else torch.zeros_like(img), coef.to(device), out, B, img[0].numel())
                img = out
            intermediates.append(img)
        return (img, intermediates) if return_intermediates else img

    @torch.no_grad()
    @_on_own_device
    def sample(self, cond, batch_size=16, use_alignment=False, alignment_kwargs=None, return_intermediates=False, x_T=None,
               verbose=False, timesteps=None, mask=None, x0=None, shape=None, return_decoded=True, **kwargs):
        """latent_diffusion.py:686-724.  Extra keywords (new API, consumed from **kwargs): sampler="ddpm"|"ddim",
        ddim_steps=50, eta=0.0, noise_tape=[x_T, n_1, ...]."""
        sampler = kwargs.pop("sampler", "ddpm")
        ddim_steps, eta = kwargs.pop("ddim_steps", 50), kwargs.pop("eta", 0.0)
        noise_tape = kwargs.pop("noise_tape", None)
        if use_alignment:
            assert self.alignment_fn is not None, "Alignment function not set."
        if shape is None:
            shape = self.get_batch_latent_shape(batch_size=batch_size)
        if self.cond_stage_model is not None:
            assert cond is not None
            if isinstance(cond, dict):
                zc = {k: (cond[k][:batch_size] if not isinstance(cond[k], list) else [v[:batch_size] for v in cond[k]]) for k in cond}
            else:
                zc = [c[:batch_size] for c in cond] if isinstance(cond, list) else cond[:batch_size]
            zc = self.cond_stage_forward(zc)
        else:
            zc = cond if isinstance(cond, torch.Tensor) else cond.get("y", None)
        y = cond if isinstance(cond, torch.Tensor) else cond.get("y", None)
        if sampler == "ddim":
            if use_alignment or mask is not None:
                raise NotImplementedError("alignment / inpainting are defined for the ancestral sampler only")
            output = self.ddim_sample_loop(zc, shape, ddim_steps=ddim_steps, eta=eta, x_T=x_T, noise_tape=noise_tape,
                                           return_intermediates=return_intermediates)
        else:
            output = self.p_sample_loop(cond=zc, shape=shape, y=y, use_alignment=use_alignment, alignment_kwargs=alignment_kwargs,
                                        return_intermediates=return_intermediates, x_T=x_T, verbose=verbose, timesteps=timesteps,
                                        mask=mask, x0=x0, noise_tape=noise_tape)
        if return_decoded:
            if return_intermediates:
                samples, inter = output
                output = [self.decode_first_stage(samples), [self.decode_first_stage(e) for e in inter]]
            else:
                output = self.decode_first_stage(output)
        return output
