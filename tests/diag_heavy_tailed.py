"""Heavy-tailed weights (prediff_amd.seeding.heavy_tailed_state_dict) through engine variants vs the oracle: which path loses what.
usage: python tests/diag_heavy_tailed.py [B]   (a diagnostic of the test tree, not a pytest module: it uses the oracle as the checker)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _templates as TP  # noqa: E402
from _cases import V1_UNET_CFG  # noqa: E402
from _weights import heavy_tailed_state_dict, seeded_input, seeded_state_dict  # noqa: E402
from oracle import unet as OU  # noqa: E402
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tmpl = TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json")
x2 = torch.cat([seeded_input("v1x", (1, 6, 16, 16, 64), 2), seeded_input("v1x2", (1, 6, 16, 16, 64), 4)])
c2 = torch.cat([seeded_input("v1c", (1, 7, 16, 16, 64), 3), seeded_input("v1c2", (1, 7, 16, 16, 64), 5)])
t2 = torch.tensor([500, 3])
rep = B // 2
xb, tb, cb = (v.repeat((rep,) + (1,) * (v.dim() - 1)).cuda() for v in (x2, t2, c2))
for kind, sd in (("gauss", seeded_state_dict(tmpl, 1234)), ("heavy", heavy_tailed_state_dict(tmpl, 77))):
    ref = OU.unet_forward(sd, V1_UNET_CFG, x2, t2, c2)
    sd16 = {k: (v.half().float() if (torch.is_floating_point(v) and v.dim() >= 2) else v) for k, v in sd.items()}
    ref16 = OU.unet_forward(sd16, V1_UNET_CFG, x2, t2, c2)
    print(f"[{kind}] oracle on fp16-rounded weights vs oracle: {rel(ref16, ref):.3e}")
    for name, prec, sw in (("fp16 pair", "fp16", {}), ("fp16 round-3 fused", "fp16", dict(fuse_pair=False)),
                           ("fp16 unfused", "fp16", dict(fuse_pair=False, fuse_attn=False, fuse_ffn=False)),
                           ("fp16x2 pair-fold", "fp16x2", dict(fuse_pair=True)), ("fp16x2 unfused", "fp16x2", dict(fuse_pair=False)),
                           ("fp16x2_lin unfused", "fp16x2_lin", dict(fuse_pair=False, fold_pair_small=False)),
                           ("fp16x2 unfused (no small pair)", "fp16x2", dict(fuse_pair=False, fold_pair_small=False)),
                           ("fp16x2, W_lo zeroed", "fp16x2", dict(fuse_pair=False, fold_pair_small=False, _zero_lo=True)),
                           ("fp32", "fp32", {})):
        net = CuboidTransformerUNet(**V1_UNET_CFG, precision=prec)
        sw = dict(sw)
        zero_lo = sw.pop("_zero_lo", False)
        for k, v in sw.items():
            setattr(net, k, v)
        net.load_state_dict(sd, strict=True)
        net = net.cuda()
        if zero_lo:        # the folded K loop with W_lo = 0: must land where the one-product engine does (same hi slabs, zeros added)
            Pk = net._ensure_packed(xb.device)          # (the key of the pack holds str(device): the forward's own device object)
            for k_, v_ in Pk.items():
                w_ = v_[0] if isinstance(v_, tuple) and len(v_) == 2 and torch.is_tensor(v_[0]) else None
                if w_ is not None and getattr(w_, "_pd_fold", False):
                    w_[w_.shape[0] // 2:].zero_()
        out = net(xb, tb, cb)[:2]
        print(f"[{kind} B={B}] {name:22s} vs oracle {rel(out, ref):.3e}   vs oracle-on-fp16-weights {rel(out, ref16):.3e}   finite {bool(torch.isfinite(out).all())}")
        del net
    # does W_lo do what it should?  (folded engine) - (the same engine with W_lo zeroed) must be  oracle(w) - oracle(GEMM weights rounded to fp16)
    GEMM_KEYS = ("in_layers.2.weight", "out_layers.3.weight", "skip_connection.weight", "qkv.weight", "proj.weight", "ffn_1.weight", "ffn_2.weight",
                 "reduction.weight", ".conv.weight", "final_proj.weight")
    sdg = {k: (v.half().float() if (torch.is_floating_point(v) and v.dim() >= 2 and any(k.endswith(e) or e in k for e in GEMM_KEYS)) else v) for k, v in sd.items()}
    refg = OU.unet_forward(sdg, V1_UNET_CFG, x2, t2, c2)
    res = {}
    for zero in (False, True):
        net = CuboidTransformerUNet(**V1_UNET_CFG, precision="fp16x2")
        net.fuse_pair, net.fold_pair_small = False, False
        net.load_state_dict(sd, strict=True)
        net = net.cuda()
        if zero:
            for k_, v_ in net._ensure_packed(xb.device).items():
                w_ = v_[0] if isinstance(v_, tuple) and len(v_) == 2 and torch.is_tensor(v_[0]) else None
                if w_ is not None and getattr(w_, "_pd_fold", False):
                    w_[w_.shape[0] // 2:].zero_()
        res[zero] = net(xb, tb, cb)[:2].double().cpu()
        del net
    print(f"[{kind} B={B}] activation-rounding term alone: folded engine vs oracle {rel(res[False], ref):.3e}; lo-zeroed engine vs oracle on GEMM-rounded weights "
          f"{rel(res[True], refg):.3e}; lo-zeroed engine vs oracle {rel(res[True], ref):.3e}")
    d_eng, d_or = res[False] - res[True], (ref - refg).double()
    cos = float((d_eng * d_or).sum() / (d_eng.norm() * d_or.norm()))
    print(f"[{kind} B={B}] effect of W_lo: engine (folded - lo zeroed) norm {float(d_eng.norm() / ref.double().norm()):.3e}, oracle (exact - GEMM weights rounded) norm "
          f"{float(d_or.norm() / ref.double().norm()):.3e}, cosine {cos:.3f}, rel diff {float((d_eng - d_or).norm() / d_or.norm()):.3f}")
    # weights that ARE fp16 numbers: W_lo = 0 exactly, so the folded engine computes what the fp16 engine computes (up to kernel choice /
    # summation order) -- how far apart the two land is the amplification of fp32 round-off by this network, not a property of the fold
    outs = {}
    for name, prec, sw in (("fp16 unfused", "fp16", dict(fuse_pair=False, fuse_attn=False, fuse_ffn=False)), ("fp16x2 unfused", "fp16x2", dict(fuse_pair=False)),
                           ("fp16 pair", "fp16", {}), ("fp16x2 pair-fold", "fp16x2", dict(fuse_pair=True))):
        net = CuboidTransformerUNet(**V1_UNET_CFG, precision=prec)
        for k, v in sw.items():
            setattr(net, k, v)
        net.load_state_dict(sd16, strict=True)
        outs[name] = net.cuda()(xb, tb, cb)[:2]
        del net
    print(f"[{kind} B={B}] fp16-representable weights: fp16x2 unfused vs fp16 unfused {rel(outs['fp16x2 unfused'], outs['fp16 unfused']):.3e}; "
          f"fp16x2 pair-fold vs fp16 pair {rel(outs['fp16x2 pair-fold'], outs['fp16 pair']):.3e}; fp16 unfused vs fp16 pair {rel(outs['fp16 unfused'], outs['fp16 pair']):.3e}; "
          f"each vs oracle16: " + ", ".join(f"{k} {rel(v, ref16):.3e}" for k, v in outs.items()))
