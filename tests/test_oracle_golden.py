"""CPU: pin the oracle (oracle/) against the golden vectors captured from the reference.

The reference has no tests of its own; tests/golden/*.npz were produced by
tests/golden/gen_golden.py importing /root/reference in the build container.
"""
import zlib

import numpy as np
import pytest
import torch

import _templates as TP
from _cases import (ATTN_CASES, MASK_CASES, REORDER_CASES, RESBLOCK3D_CASES, TINY_UNET_CFGS, TINY_VAE_CFG,
                    V1_UNET_CFG, V1_VAE_CFG)
from _weights import seeded_input, seeded_state_dict
from oracle import diffusion as OD
from oracle import unet as OU
from oracle import vae as OV



@pytest.fixture(autouse=True)
def _no_grad():
    """the oracle runs without autograd -- per test, not process-wide (other modules of the same pytest process take gradients)"""
    with torch.no_grad():
        yield


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ------------------------------------------------------------------ integer / index work: bit exact
@pytest.mark.parametrize("i", range(len(REORDER_CASES)))
def test_reorder_ids(golden, i):
    shape, cuboid, strategy = REORDER_CASES[i]
    ids = OU.cuboid_token_ids(shape, cuboid, strategy)
    assert np.array_equal(ids.numpy(), golden("cuboid_index")[f"reorder_{i}"])
    # reverse . reorder == id
    x = torch.randn(2, *shape, 3)
    assert torch.equal(OU.cuboid_reorder_reverse(OU.cuboid_reorder(x, cuboid, strategy), cuboid, strategy, shape), x)


@pytest.mark.parametrize("i", range(len(MASK_CASES)))
def test_attention_mask(golden, i):
    shape, cuboid, shift, strategy, padding_type = MASK_CASES[i]
    g = golden("cuboid_index")
    cub, sh = OU.clamp_cuboid(shape, cuboid, shift, strategy)
    assert list(cub) + list(sh) == g[f"mask_{i}_clamped"].tolist()
    m = OU.cuboid_attention_mask(shape, cub, sh, strategy, padding_type)
    assert np.array_equal(m.numpy(), g[f"mask_{i}"])


def test_mask_all_true_without_pad_or_shift():
    m = OU.cuboid_attention_mask((13, 16, 16), (13, 1, 1), (0, 0, 0), ("l", "l", "l"), "zeros")
    assert bool(m.all())


@pytest.mark.parametrize("i", range(len(ATTN_CASES)))
def test_relative_position_index(golden, i):
    c = ATTN_CASES[i]
    assert np.array_equal(OU.relative_position_index(c["cuboid"]).numpy(), golden("attn_layer")[f"relidx_{i}"])


# ------------------------------------------------------------------ floating point layers: <= 1e-5 rel-L2
@pytest.mark.parametrize("i", range(len(ATTN_CASES)))
def test_attention_layer(golden, i):
    c = ATTN_CASES[i]
    sd = seeded_state_dict(TP.attn_layer(c["dim"], c["heads"], c["cuboid"]), 100 + i)
    x = seeded_input(f"attn{i}", (c["B"],) + tuple(c["shape"]) + (c["dim"],), 1)
    y = OU.cuboid_self_attention(sd, "", x, c["heads"], c["cuboid"], c["shift"], c["strategy"], c["padding_type"])
    g = golden("attn_layer")
    if c["dim"] >= 256:
        assert rel_l2(y[:, :, ::2, ::2, ::4], g[f"y_{i}_slice"]) < 1e-5
        assert abs(float(y.double().abs().sum()) / float(g[f"y_{i}_abs_sum"][0]) - 1) < 1e-5
    else:
        assert rel_l2(y, g[f"y_{i}"]) < 1e-5


def test_small_layers(golden):
    g = golden("small_layers")
    for i, (act, gated) in enumerate([("gelu", False), ("leaky", False), ("gelu", True)]):
        sd = seeded_state_dict(TP.ffn(32, 128, gated), 200 + i)
        x = seeded_input(f"ffn{i}", (2, 3, 4, 4, 32), 1)
        assert rel_l2(OU.positionwise_ffn(sd, "", x, act, gated), g[f"ffn_{i}"]) < 1e-5
    for i, (shape, ptype) in enumerate([((3, 8, 8), "zeros"), ((3, 7, 6), "zeros"), ((3, 7, 6), "nearest")]):
        sd = seeded_state_dict(TP.patch_merge(16, 32), 210 + i)
        x = seeded_input(f"pm{i}", (2,) + shape + (16,), 1)
        assert rel_l2(OU.patch_merging_3d(sd, "", x, (1, 2, 2), ptype), g[f"pm_{i}"]) < 1e-5
    sd = seeded_state_dict(TP.upsample3d(32, 16), 220)
    assert rel_l2(OU.upsample_3d(sd, "", seeded_input("up0", (2, 3, 4, 4, 32), 1), (3, 8, 8)), g["up_0"]) < 1e-5
    sd = seeded_state_dict(TP.pos_embed(16, 5, 8, 8), 230)
    assert rel_l2(OU.pos_embed(sd, "", seeded_input("pos0", (2, 5, 8, 8, 16), 1)), g["pos_0"]) < 1e-6
    t = torch.as_tensor(g["temb_t"])
    assert rel_l2(OU.timestep_embedding(t, 64), g["temb_64"]) < 1e-6
    assert rel_l2(OU.timestep_embedding(t, 33), g["temb_33"]) < 1e-6
    sd = seeded_state_dict(TP.time_embed_layer(64, 256), 240)
    import torch.nn.functional as F
    y = F.linear(F.silu(F.linear(torch.as_tensor(g["temb_64"]), sd["layer.0.weight"], sd["layer.0.bias"])),
                 sd["layer.2.weight"], sd["layer.2.bias"])
    assert rel_l2(y, g["tel_0"]) < 1e-5


@pytest.mark.parametrize("i", range(len(RESBLOCK3D_CASES)))
def test_time_embed_resblock(golden, i):
    c = RESBLOCK3D_CASES[i]
    sd = seeded_state_dict(TP.resblock3d(c["cin"], c["cout"], c["emb"], c["ssn"]), 300 + i)
    x = seeded_input(f"rb{i}", (2, c["cin"]) + tuple(c["shape"]), 1).permute(0, 2, 3, 4, 1)
    emb = seeded_input(f"rbe{i}", (2, c["emb"]), 1) if c["emb"] is not None else None
    y = OU.time_embed_resblock(sd, "", x, emb, c["ssn"])
    assert rel_l2(y, golden("resblock3d")[f"y_{i}"]) < 1e-5


# ------------------------------------------------------------------ whole denoiser
@pytest.mark.parametrize("name", list(TINY_UNET_CFGS))
def test_tiny_unet(golden, name):
    cfg = TINY_UNET_CFGS[name]
    sd = seeded_state_dict(TP.unet_template(cfg, "tiny_unet_schema.json", name), 400 + zlib.crc32(name.encode()) % 97)
    x = seeded_input(name + "x", (2,) + tuple(cfg["target_shape"]), 2)
    cond = seeded_input(name + "c", (2,) + tuple(cfg["input_shape"]), 3)
    out = OU.unet_forward(sd, cfg, x, torch.tensor([7, 431]), cond)
    assert rel_l2(out, golden("tiny_unet")[f"{name}_out"]) < 1e-4


def test_v1_unet_full_size(golden):
    """v1 config (136.8 M params): ONE forward, ~1-2 s on CPU; <= 1e-4 rel-L2."""
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    assert len(sd) == 688
    x = seeded_input("v1x", (1, 6, 16, 16, 64), 2)
    cond = seeded_input("v1c", (1, 7, 16, 16, 64), 3)
    out = OU.unet_forward(sd, V1_UNET_CFG, x, torch.tensor([500]), cond)
    g = golden("v1_unet")
    assert rel_l2(out[0, :, ::4, ::4, ::8], g["out_slice"]) < 1e-4
    assert abs(float(out.double().abs().sum()) / float(g["out_abs_sum"][0]) - 1) < 1e-4
    assert rel_l2(out, g["out_full_f16"].astype(np.float32)) < 2e-3      # fp16 storage of the full tensor


# ------------------------------------------------------------------ VAE
def test_vae_layers(golden):
    g = golden("vae")
    sd = seeded_state_dict(TP.resnet2d(32, 32), 500)
    assert rel_l2(OV.resnet_block_2d(sd, "", seeded_input("vrb0", (2, 32, 8, 8), 1), 8), g["rb_same"]) < 1e-5
    sd = seeded_state_dict(TP.resnet2d(32, 64), 501)
    assert rel_l2(OV.resnet_block_2d(sd, "", seeded_input("vrb1", (2, 32, 8, 8), 1), 8), g["rb_diff"]) < 1e-5
    import torch.nn.functional as F
    sd = seeded_state_dict(TP.conv2d(32, 32), 502)
    y = F.conv2d(F.pad(seeded_input("vds", (2, 32, 8, 8), 1), (0, 1, 0, 1)), sd["conv.weight"], sd["conv.bias"], stride=2)
    assert rel_l2(y, g["down"]) < 1e-5
    sd = seeded_state_dict(TP.conv2d(32, 32), 503)
    y = F.conv2d(F.interpolate(seeded_input("vus", (2, 32, 4, 4), 1), scale_factor=2.0), sd["conv.weight"],
                 sd["conv.bias"], padding=1)
    assert rel_l2(y, g["up"]) < 1e-5
    sd = seeded_state_dict(TP.vae_attention(64), 504)
    assert rel_l2(OV.attention_block(sd, "", seeded_input("vat", (2, 64, 4, 4), 1), 8), g["attn"]) < 1e-5
    mode, logvar = OV.gaussian_mode_and_logvar(seeded_input("vmom", (2, 8, 4, 4), 1) * 25.0)
    assert np.array_equal(mode.numpy(), g["dist_mode"]) and np.array_equal(logvar.numpy(), g["dist_logvar"])
    assert float(logvar.max()) == 20.0 and float(logvar.min()) == -30.0


def test_tiny_vae(golden):
    g = golden("vae")
    sd = seeded_state_dict(TP.from_schema("tiny_vae_schema.json"), 510)
    x = seeded_input("vaex", (3, 1, 32, 32), 1, kind="uniform")
    mom = OV.vae_encode_moments(sd, TINY_VAE_CFG, x)
    assert rel_l2(mom, g["tiny_moments"]) < 1e-5
    assert rel_l2(OV.vae_encode_mode(sd, TINY_VAE_CFG, x), g["tiny_mode"]) < 1e-5
    z = seeded_input("vaez", (3, 4, 8, 8), 1)
    assert rel_l2(OV.vae_decode(sd, TINY_VAE_CFG, z), g["tiny_dec"]) < 1e-5


def test_v1_vae_full_size(golden):
    g = golden("v1_vae")
    sd = seeded_state_dict(TP.from_schema("v1_vae_schema.json"), 4321)
    assert len(sd) == 248
    mode = OV.vae_encode_mode(sd, V1_VAE_CFG, seeded_input("v1vaex", (1, 1, 128, 128), 1, kind="uniform"))
    assert abs(float(mode.double().abs().sum()) / float(g["mode_abs_sum"][0]) - 1) < 1e-4
    assert rel_l2(mode, g["mode_f16"].astype(np.float32)) < 2e-3
    dec = OV.vae_decode(sd, V1_VAE_CFG, seeded_input("v1vaez", (1, 64, 16, 16), 1))
    assert abs(float(dec.double().abs().sum()) / float(g["dec_abs_sum"][0]) - 1) < 1e-4
    assert rel_l2(dec, g["dec_f16"].astype(np.float32)) < 2e-3


# ------------------------------------------------------------------ diffusion
def test_schedule_buffers(golden):
    g = golden("schedule")
    buf = OD.schedule_buffers(OD.beta_schedule("linear", 1000))
    for k, v in buf.items():
        assert np.array_equal(v, g[k]), k                      # float64 math -> fp32: bit exact
    for s in ("cosine", "sqrt_linear", "sqrt"):
        assert np.allclose(OD.beta_schedule(s, 1000), g["betas_" + s], rtol=1e-12, atol=0)
    # known answers recorded in SURVEY.md §8(a) a4
    b = OD.beta_schedule("linear", 1000)
    assert b[0] == pytest.approx(1e-4, rel=1e-12) and b[999] == pytest.approx(0.02, rel=1e-12)
    assert b[1] == pytest.approx(1.0264836435083402e-4, rel=1e-12)
    assert float(np.cumprod(1 - b)[499]) == pytest.approx(0.3331877673563, rel=1e-9)
    assert buf["posterior_log_variance_clipped"][0] == np.float32(np.log(1e-20))


def test_ddim_helpers(golden):
    g = golden("schedule")
    ac = np.cumprod(1.0 - OD.beta_schedule("linear", 1000)).astype(np.float32).astype(np.float64)
    for S in (10, 50, 100):
        steps = OD.ddim_timesteps(S, 1000)
        assert np.array_equal(steps, g[f"ddim_steps_{S}"])
        for eta in (0.0, 1.0):
            sig, a, ap = OD.ddim_sampling_parameters(ac, np.minimum(steps, 999), eta)
            assert np.allclose(sig, g[f"ddim_sigma_{S}_{int(eta)}"], rtol=1e-12, atol=0)
            assert np.allclose(a, g[f"ddim_a_{S}"], rtol=1e-12) and np.allclose(ap, g[f"ddim_aprev_{S}"], rtol=1e-12)
    assert np.array_equal(OD.ddim_timesteps(20, 1000, "quad"), g["ddim_steps_quad_20"])
    assert OD.ddim_timesteps(50, 1000)[[0, 1, -1]].tolist() == [1, 21, 981]


def _tiny_ldm_state():
    cfg = TINY_UNET_CFGS["axial"]
    sd = seeded_state_dict(TP.unet_template(cfg, "tiny_unet_schema.json", "axial"), 600)
    vsd = seeded_state_dict(TP.from_schema("tiny_vae_schema.json"), 601)
    buf = {k: torch.as_tensor(v) for k, v in OD.schedule_buffers(OD.beta_schedule("linear", 1000)).items()}
    return cfg, sd, vsd, buf


def test_p_sample(golden):
    g = golden("p_sample")
    cfg, sd, _, buf = _tiny_ldm_state()
    B = 2
    zc = seeded_input("dzc", (B,) + tuple(cfg["input_shape"]), 5)
    zt = seeded_input("dzt", (B,) + tuple(cfg["target_shape"]), 6)
    for tt in (999, 500, 1, 0):
        t = torch.full((B,), tt, dtype=torch.long)
        eps = OU.unet_forward(sd, cfg, zt, t, zc)
        assert rel_l2(eps, g[f"eps_{tt}"]) < 1e-4
        out = OD.ddpm_step(buf, zt, torch.as_tensor(g[f"eps_{tt}"]), t, torch.as_tensor(g[f"psample_noise_{tt}"]))
        assert rel_l2(out, g[f"psample_{tt}"]) < 1e-6
    # t == 0 adds no noise (latent_diffusion.py:624-631)
    t0 = torch.zeros(B, dtype=torch.long)
    e = torch.as_tensor(g["eps_0"])
    assert torch.equal(OD.ddpm_step(buf, zt, e, t0, torch.randn_like(zt)), OD.ddpm_step(buf, zt, e, t0, torch.zeros_like(zt)))


def test_sample_loop(golden):
    """LatentDiffusion.sample(timesteps=3): VAE-encode ctx -> 3 ancestral steps on the noise tape -> decode."""
    g = golden("sample3")
    cfg, sd, vsd, buf = _tiny_ldm_state()
    B, T_in = 2, cfg["input_shape"][0]
    y = seeded_input("dy", (B, T_in, 32, 32, 1), 8, kind="uniform")
    frames = y.permute(0, 1, 4, 2, 3).reshape(B * T_in, 1, 32, 32)                   # "(N T) C H W"
    zc = OV.vae_encode_mode(vsd, TINY_VAE_CFG, frames)
    zc = zc.reshape(B, T_in, *zc.shape[1:]).permute(0, 1, 3, 4, 2)                    # back to N T H W C
    assert rel_l2(zc, g["zc"]) < 1e-5
    tape = [torch.as_tensor(v) for v in g["tape"]]
    traj = OD.ddpm_sample_loop(buf, lambda z, t, c: OU.unet_forward(sd, cfg, z, t, c), zc, tape, 3)
    assert rel_l2(traj[-1], g["latent"]) < 1e-4
    z0 = traj[-1]
    T_out = z0.shape[1]
    dec = OV.vae_decode(vsd, TINY_VAE_CFG, z0.permute(0, 1, 4, 2, 3).reshape(B * T_out, -1, *z0.shape[2:4]))
    dec = dec.reshape(B, T_out, 1, 32, 32).permute(0, 1, 3, 4, 2)
    assert rel_l2(dec, g["decoded"]) < 1e-4


def test_cond_schedule_loop(golden):
    """shorten_cond_schedule (num_timesteps_cond = 4): cond_ids bit-equal, three ancestral steps with the condition re-noised in front of
    each one on the recorded tapes (reference latent_diffusion.py:295-299, 665-667)."""
    g = golden("cond_schedule")
    cfg, sd, _, buf = _tiny_ldm_state()
    ids = OD.cond_schedule(1000, 4)
    assert ids.dtype == g["cond_ids"].dtype and np.array_equal(ids, g["cond_ids"])
    zc = seeded_input("dzc", (2,) + tuple(cfg["input_shape"]), 5)
    traj = OD.ddpm_sample_loop(buf, lambda z, t, c: OU.unet_forward(sd, cfg, z, t, c), zc, [torch.as_tensor(v) for v in g["tape_x"]], 3,
                               cond_ids=ids, cond_tape=[torch.as_tensor(v) for v in g["tape_c"]])
    assert rel_l2(traj[-1], g["latent"]) < 1e-4
    plain = OD.ddpm_sample_loop(buf, lambda z, t, c: OU.unet_forward(sd, cfg, z, t, c), zc, [torch.as_tensor(v) for v in g["tape_x"]], 3)
    assert rel_l2(plain[-1], g["latent"]) > 1e-2          # the option matters on this input


def test_ddim_self_consistency():
    """PARITY-UNPINNED DDIM rule: eta=0 is deterministic; z0 formula equals predict_start_from_noise."""
    buf = {k: torch.as_tensor(v) for k, v in OD.schedule_buffers(OD.beta_schedule("linear", 1000)).items()}
    zt, eps = torch.randn(2, 2, 4, 4, 3), torch.randn(2, 2, 4, 4, 3)
    t = torch.tensor([500, 500])
    a_t = buf["alphas_cumprod"][t]
    a_prev = buf["alphas_cumprod"][t - 20]
    sig0 = torch.zeros(2)
    a = OD.ddim_step(zt, eps, a_t, a_prev, sig0, torch.randn_like(zt))
    b = OD.ddim_step(zt, eps, a_t, a_prev, sig0, torch.randn_like(zt))
    assert torch.equal(a, b)
    z0_ddpm = buf["sqrt_recip_alphas_cumprod"][500] * zt - buf["sqrt_recipm1_alphas_cumprod"][500] * eps
    z0_ddim = (zt - (1 - a_t[0]).sqrt() * eps) / a_t[0].sqrt()
    assert rel_l2(z0_ddim, z0_ddpm) < 1e-5
    # with a_prev == a_t and sigma == 0 the step is the identity
    same = OD.ddim_step(zt, eps, a_t, a_t, sig0, torch.zeros_like(zt))
    assert rel_l2(same, zt) < 1e-5


# ------------------------------------------------------------------ BASELINE config 1 stand-in and config 4 at full size
def test_nbody_standin(golden):
    """Config 1 ("N-body MNIST 64x64, 10-step DDIM, 1 sample, CPU"): the oracle's VAE + denoiser + DDIM loop against the run of the
    reference's own modules at the stand-in sizes (tests/golden/gen_golden.py:gen_nbody)."""
    from _cases import NBODY_UNET_CFG, NBODY_VAE_CFG
    g = golden("nbody")
    usd = seeded_state_dict(TP.unet_template(NBODY_UNET_CFG, "nbody_schema.json", "unet"), 800)
    vsd = seeded_state_dict(TP.from_schema("nbody_schema.json", "vae"), 801)
    y = seeded_input("nby", (1, 10, 64, 64, 1), 0, kind="uniform")
    frames = y.permute(0, 1, 4, 2, 3).reshape(10, 1, 64, 64)
    zc = OV.vae_encode_mode(vsd, NBODY_VAE_CFG, frames).reshape(1, 10, 4, 16, 16).permute(0, 1, 3, 4, 2)
    assert rel_l2(zc, g["zc"]) < 1e-5
    ac = np.cumprod(1.0 - OD.beta_schedule("linear", 1000)).astype(np.float32)
    xT = seeded_input("nbxT", (1, 10, 16, 16, 4), 1)
    tape = [xT] + [torch.zeros_like(xT)] * 10
    lat = OD.ddim_sample_loop(ac, lambda z, t, c: OU.unet_forward(usd, NBODY_UNET_CFG, z, t, c), zc, tape, 10, eta=0.0)[-1]
    assert rel_l2(lat, g["latent"]) < 1e-4
    dec = OV.vae_decode(vsd, NBODY_VAE_CFG, lat.permute(0, 1, 4, 2, 3).reshape(10, 4, 16, 16))
    dec = dec.reshape(1, 10, 1, 64, 64).permute(0, 1, 3, 4, 2)
    assert rel_l2(dec, g["decoded"]) < 1e-4


def test_v1_aligned_step(golden):
    """Config 4 at full size: oracle denoiser + prediff_amd.alignment guidance (PyTorch autograd, CPU) + oracle step epilogue against
    the reference's knowledge-aligned p_sample at t in {99, 0}."""
    from _cases import V1_ALIGN_ARGS
    from prediff_amd.alignment import SEVIRAvgIntensityAlignment
    g = golden("v1_aligned")
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    al = SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid", model_args=dict(V1_ALIGN_ARGS))
    al.model.load_state_dict(seeded_state_dict(al.model.state_dict(), 701))
    buf = {k: torch.as_tensor(v) for k, v in OD.schedule_buffers(OD.beta_schedule("linear", 1000)).items()}
    B = 2
    zt, zc = seeded_input("v1azt", (B, 6, 16, 16, 64), 12), seeded_input("v1azc", (B, 7, 16, 16, 64), 13)
    avg = torch.as_tensor(g["avg_x_gt"])
    for tt in (99, 0):
        t = torch.full((B,), tt, dtype=torch.long)
        shift = al.get_mean_shift(zt, t, y=None, zc=zc, avg_x_gt=avg)
        assert rel_l2(shift[:, :, ::2, ::2, ::4], g[f"shift_{tt}_slice"]) < 1e-4
        with torch.no_grad():
            out = OD.ddpm_step(buf, zt, OU.unet_forward(sd, V1_UNET_CFG, zt, t, zc), t, seeded_input(f"v1an{tt}", (B, 6, 16, 16, 64), 14),
                               mean_shift=shift)
        assert rel_l2(out[:, :, ::2, ::2, ::4], g[f"out_{tt}_slice"]) < 1e-4
        assert abs(float(out.double().abs().sum()) / float(g[f"out_{tt}_abs_sum"][0]) - 1) < 1e-4
