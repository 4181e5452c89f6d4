"""Wall-clock split of LatentDiffusion.sample() at the v1 configuration (run on the GPU box):
context encode (VAE encoder, 7 frames / trajectory) | DDIM-50 loop | decode (VAE decoder, 6 frames / trajectory)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prediff_amd.presets import V1_UNET_CFG, V1_VAE_CFG
from prediff_amd.seeding import seeded_state_dict
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
from prediff_amd.autoencoder_kl import AutoencoderKL
from prediff_amd.latent_diffusion import LatentDiffusion

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
net = CuboidTransformerUNet(**V1_UNET_CFG, precision="bf16")
net.load_state_dict(seeded_state_dict(net.state_dict(), 1234))
vae = AutoencoderKL(**V1_VAE_CFG, precision="bf16")
vae.load_state_dict(seeded_state_dict(vae.state_dict(), 77))
ldm = LatentDiffusion(torch_nn_module=net, layout="NTHWC", data_shape=(6, 128, 128, 1), timesteps=1000, beta_schedule="linear",
                      use_ema=False, latent_shape=(6, 16, 16, 64), first_stage_model=vae, cond_stage_model="__is_first_stage__",
                      scale_factor=1.0).to(dev).eval()
ctx = torch.rand(B, 7, 128, 128, 1, device=dev)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t0


with torch.no_grad():
    ldm.sample(cond={"y": ctx}, batch_size=B, sampler="ddim", ddim_steps=2)          # warm-up (workspaces, graph)
    out, t_all = timed(lambda: ldm.sample(cond={"y": ctx}, batch_size=B, sampler="ddim", ddim_steps=50))
    zc, t_enc = timed(lambda: ldm.cond_stage_forward({"y": ctx}))
    z = torch.randn(ldm.get_batch_latent_shape(B), device=dev)
    _, t_dec = timed(lambda: ldm.decode_first_stage(z))
print(f"B={B}: sample() DDIM-50 total {t_all * 1e3:.1f} ms; encode ctx {t_enc * 1e3:.1f} ms; decode {t_dec * 1e3:.1f} ms; "
      f"loop ~{(t_all - t_enc - t_dec) * 1e3:.1f} ms; out {tuple(out.shape)} finite={bool(torch.isfinite(out).all())}")
