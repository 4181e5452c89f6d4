"""GPU: frame-wise KL-VAE (prediff_amd.AutoencoderKL) against the golden vectors captured from the reference and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _templates as TP  # noqa: E402
from _cases import TINY_VAE_CFG, V1_VAE_CFG  # noqa: E402
from _weights import seeded_input, seeded_state_dict  # noqa: E402
from oracle import vae as OV  # noqa: E402
from prediff_amd.autoencoder_kl import AutoencoderKL  # noqa: E402
from prediff_amd.distributions import DiagonalGaussianDistribution  # noqa: E402

TOL = {"fp32": 1e-4, "bf16": 3e-2, "fp16": 4e-3}       # fp16: IEEE-half operands (8x finer than bf16; measured bf16 ~1e-2)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_tiny_vae(golden, precision):
    g = golden("vae")
    sd = seeded_state_dict(TP.from_schema("tiny_vae_schema.json"), 510)
    vae = AutoencoderKL(**TINY_VAE_CFG, precision=precision)
    vae.load_state_dict(sd, strict=True)
    vae = vae.cuda()
    x = seeded_input("vaex", (3, 1, 32, 32), 1, kind="uniform")
    post = vae.encode(x.cuda())
    assert isinstance(post, DiagonalGaussianDistribution)
    e1, e2 = rel_l2(post.parameters, g["tiny_moments"]), rel_l2(post.mode(), g["tiny_mode"])
    z = seeded_input("vaez", (3, 4, 8, 8), 1)
    e3 = rel_l2(vae.decode(z.cuda()), g["tiny_dec"])
    print(f"[tiny vae {precision}] moments {e1:.3e} mode {e2:.3e} decode {e3:.3e}")
    assert max(e1, e2, e3) < TOL[precision]
    # sliced decoding gives the same frames
    vae.enable_slicing()
    assert rel_l2(vae.decode(z.cuda()), g["tiny_dec"]) < TOL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_v1_vae_full_size(golden, precision):
    g = golden("v1_vae")
    sd = seeded_state_dict(TP.from_schema("v1_vae_schema.json"), 4321)
    vae = AutoencoderKL(**V1_VAE_CFG, precision=precision)
    vae.load_state_dict(sd, strict=True)
    vae = vae.cuda()
    x = seeded_input("v1vaex", (1, 1, 128, 128), 1, kind="uniform")
    z = seeded_input("v1vaez", (1, 64, 16, 16), 1)
    mode = vae.encode(x.cuda()).mode()
    dec = vae.decode(z.cuda())
    ref_mode = OV.vae_encode_mode(sd, V1_VAE_CFG, x)
    ref_dec = OV.vae_decode(sd, V1_VAE_CFG, z)
    e1, e2 = rel_l2(mode, ref_mode), rel_l2(dec, ref_dec)
    print(f"[v1 vae {precision}] encode-mode vs oracle {e1:.3e} (vs reference f16 {rel_l2(mode, g['mode_f16'].astype(np.float32)):.3e}), "
          f"decode vs oracle {e2:.3e} (vs reference f16 {rel_l2(dec, g['dec_f16'].astype(np.float32)):.3e})")
    assert e1 < TOL[precision] and e2 < TOL[precision]
    # batch of frames == frame by frame (7 context frames of one sample)
    xs = seeded_input("v1vaexs", (7, 1, 128, 128), 2, kind="uniform").cuda()
    m7 = vae.encode(xs).mode()
    assert rel_l2(m7[3:4], vae.encode(xs[3:4]).mode()) < 1e-6
