"""SURVEY §8 f4 -- what the reference's training side computes without a gradient, against goldens recorded from the reference
(tests/golden/gen_golden.py:gen_training_side): q_sample, the variational-bound weights, the loss of a batch (p_losses), the EMA
shadow weights (utils/ema.py) and the with-EMA evaluation of validation_step (latent_diffusion.py:280-293, 487-551).
CPU: schedule weights, q_sample, LitEma arithmetic, checkpoint key schema.  GPU: the losses through the HIP denoiser."""
import json
import os

import numpy as np
import pytest
import torch

from _cases import TINY_UNET_CFGS
from _weights import seeded_input, seeded_state_dict
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
from prediff_amd.ema import LitEma
from prediff_amd.latent_diffusion import LatentDiffusion

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = TINY_UNET_CFGS["axial"]


def _ldm(precision="fp32", use_ema=True):
    net = CuboidTransformerUNet(**CFG, precision=precision)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 600))
    ldm = LatentDiffusion(torch_nn_module=net, layout="NTHWC", data_shape=(CFG["target_shape"][0], 32, 32, 1), timesteps=1000,
                          beta_schedule="linear", use_ema=use_ema, original_elbo_weight=0.1, latent_shape=tuple(CFG["target_shape"]),
                          first_stage_model=None, cond_stage_model=None, scale_factor=1.0)
    return ldm.eval()


def _inputs():
    B = 3
    lat = (B,) + tuple(CFG["target_shape"])
    return (seeded_input("tsx0", lat, 31), seeded_input("tsn", lat, 32), seeded_input("tszc", (B,) + tuple(CFG["input_shape"]), 33),
            torch.tensor([999, 417, 0]))


def _move_and_track(net, ema):
    """the recorded sequence: three steps of p <- 0.9 p + 0.01, an EMA update after each"""
    for _ in range(3):
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(0.9).add_(0.01)
        ema(net)


def test_lvlb_weights_and_q_sample_bit_exact(golden):
    g = golden("train_side")
    ldm = _ldm()
    x0, noise, _, t = _inputs()
    assert np.array_equal(ldm.lvlb_weights.numpy(), g["lvlb_weights"])
    assert "lvlb_weights" not in ldm.state_dict()                      # non-persistent, as in the reference
    assert np.array_equal(ldm.q_sample(x0, t, noise).numpy(), g["q_sample"])


def test_lit_ema_update_rule_and_schema(golden):
    g = golden("train_side")
    probe = json.load(open(os.path.join(HERE, "golden", "train_side_probe.json")))
    ldm = _ldm()
    net = ldm.torch_nn_module
    assert not any(p.requires_grad for p in net.parameters())          # the engine freezes its parameters: tracked all the same
    ema = ldm.model_ema
    assert isinstance(ema, LitEma) and len(ema.m_name2s_name) == probe["n_params"]
    # checkpoint schema of the reference: model_ema.decay, model_ema.num_updates, one buffer per parameter named without the dots
    keys = {k for k in ldm.state_dict() if k.startswith("model_ema.")}
    assert keys == {"model_ema.decay", "model_ema.num_updates"} | {"model_ema." + n.replace(".", "") for n, _ in net.named_parameters()}
    assert float(ema.decay) == np.float32(0.9999) and int(ema.num_updates) == 0
    _move_and_track(net, ema)
    assert int(ema.num_updates) == int(g["ema_num_updates"][0]) == 3
    shadows = dict(ema.named_buffers())
    for i, n in enumerate(probe["probe"]):
        assert np.array_equal(shadows[ema.m_name2s_name[n]].numpy(), g[f"ema_shadow_{i}"]), n
    tot = sum(float(v.double().abs().sum()) for k, v in shadows.items() if k not in ("decay", "num_updates"))
    assert abs(tot / float(g["ema_abs_sum"][0]) - 1) < 1e-12
    # store / copy_to / restore round trip; copy_to bumps the parameter versions (the HIP engine re-packs on that)
    before = [p.detach().clone() for p in net.parameters()]
    v0 = next(net.parameters())._version
    with ldm.ema_scope():
        assert next(net.parameters())._version > v0
        assert all(torch.equal(p, shadows[ema.m_name2s_name[n]]) for n, p in net.named_parameters())
    assert all(torch.equal(p, b) for p, b in zip(net.parameters(), before))
    # constant-decay form and the reference's own tracking rule
    lin = torch.nn.Linear(3, 2)
    e2 = LitEma(lin, decay=0.5, use_num_upates=False)
    assert int(e2.num_updates) == -1 and set(e2.m_name2s_name) == {"weight", "bias"}
    with torch.no_grad():
        lin.weight.add_(1.0)
    w_shadow = e2.weight.clone()
    e2(lin)
    assert torch.allclose(e2.weight, w_shadow + 0.5 * (lin.weight - w_shadow))
    lin.bias.requires_grad_(False)
    assert set(LitEma(lin).m_name2s_name) == {"weight"}                # mixed module: requires_grad decides, as in the reference
    with pytest.raises(ValueError):
        LitEma(lin, decay=1.5)


def test_training_step_is_refused():
    ldm = _ldm(use_ema=False)
    assert not hasattr(ldm, "model_ema")
    with pytest.raises(NotImplementedError):
        ldm.training_step({}, 0)
    ldm.on_train_batch_end()                                            # no EMA: a no-op
    with ldm.ema_scope():
        pass


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_p_losses_and_ema_evaluation_vs_reference(golden, precision, tol):
    """The loss of a batch through the HIP denoiser, with the current weights, with the EMA weights swapped in (ema_scope) and after
    the restore, against the reference's values on the same inputs."""
    g = golden("train_side")
    ldm = _ldm(precision).cuda()
    x0, noise, zc, t = (v.cuda() for v in _inputs())
    loss, ld = ldm.p_losses(x0, zc, t, noise=noise)
    assert set(ld) == {"val/loss_simple", "val/loss_vlb", "val/loss"} and not loss.requires_grad
    for k, v in ld.items():
        ref = float(g["ld_" + k.replace("/", "_")][0])
        assert abs(float(v) / ref - 1) < tol, (k, float(v), ref)
    assert abs(float(loss) / float(g["loss"][0]) - 1) < tol
    _move_and_track(ldm.torch_nn_module, ldm.model_ema)
    with ldm.ema_scope():
        loss_e, _ = ldm.p_losses(x0, zc, t, noise=noise)
    loss_m, _ = ldm.p_losses(x0, zc, t, noise=noise)
    print(f"[p_losses {precision}] loss {float(loss):.6f} (ref {float(g['loss'][0]):.6f}), with EMA weights {float(loss_e):.6f} "
          f"(ref {float(g['loss_ema'][0]):.6f}), moved weights {float(loss_m):.6f} (ref {float(g['loss_moved'][0]):.6f})")
    assert abs(float(loss_e) / float(g["loss_ema"][0]) - 1) < tol
    assert abs(float(loss_m) / float(g["loss_moved"][0]) - 1) < tol

    class _Batch(LatentDiffusion):       # the dataset-dependent piece a script supplies (train_sevirlr_prediff.py:733-759)
        def get_input(self, batch, **kwargs):
            return batch["z"], {"y": batch["zc"]}
    ldm.__class__ = _Batch
    torch.manual_seed(0)
    out = ldm.validation_step({"z": x0, "zc": zc}, 0)
    assert set(out) == {"val/loss_simple", "val/loss_vlb", "val/loss", "val/loss_simple_ema", "val/loss_vlb_ema", "val/loss_ema"}
    assert all(bool(torch.isfinite(v)) for v in out.values())
