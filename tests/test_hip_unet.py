"""GPU: the HIP denoiser (prediff_amd.CuboidTransformerUNet) against the oracle and the committed golden outputs.

Tolerances: precision="fp32" (bf16 hi/lo split GEMMs, fp32 attention) <= 1e-4 rel-L2 per forward vs the
CPU oracle; precision="bf16" (throughput mode) is reported and bounded at 2e-2 per forward -- the reference itself is
fp32-only, bf16 is an engine choice.
"""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _templates as TP  # noqa: E402
from _cases import TINY_UNET_CFGS, V1_UNET_CFG  # noqa: E402
from _weights import heavy_tailed_state_dict, seeded_input, seeded_state_dict  # noqa: E402
from oracle import unet as OU  # noqa: E402
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet  # noqa: E402

TOL = {"fp32": 1e-4, "bf16": 2e-2, "fp16": 2.5e-3, "fp16x2": 1.5e-3}      # fp16: IEEE-half operands, 8x finer than bf16 (measured bf16 ~7e-3 per forward); fp16x2: + exact weights


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "fp16x2"])
@pytest.mark.parametrize("name", list(TINY_UNET_CFGS))
def test_tiny_unet_vs_oracle_and_golden(golden, name, precision):
    cfg = TINY_UNET_CFGS[name]
    sd = seeded_state_dict(TP.unet_template(cfg, "tiny_unet_schema.json", name), 400 + zlib.crc32(name.encode()) % 97)
    net = CuboidTransformerUNet(**cfg, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    x = seeded_input(name + "x", (2,) + tuple(cfg["target_shape"]), 2)
    cond = seeded_input(name + "c", (2,) + tuple(cfg["input_shape"]), 3)
    t = torch.tensor([7, 431])
    if precision == "fp32" and name in ("full", "divided_st_16"):
        # cuboids of more than 64 slots run on the online-softmax MFMA core, which takes bf16 q/k/v: the fp32-class engine says so
        from prediff_amd._lib import PrediffHipError
        with pytest.raises(PrediffHipError, match="cuboid volume"):
            net(x.cuda(), t.cuda(), cond.cuda())
        return
    out = net(x.cuda(), t.cuda(), cond.cuda())
    ref = OU.unet_forward(sd, cfg, x, t, cond)
    e_or, e_gold = rel_l2(out, ref), rel_l2(out, golden("tiny_unet")[f"{name}_out"])
    print(f"[{name} {precision}] rel-L2 vs oracle {e_or:.3e}, vs reference golden {e_gold:.3e}")
    assert e_or < TOL[precision] and e_gold < TOL[precision]


_ORACLE_V1_B2 = []


def _oracle_v1_b2(sd, x2, t2, c2):
    """The oracle's CPU forward of the B = 2 case, once for the three precision parametrisations (same seeded weights and inputs)."""
    if not _ORACLE_V1_B2:
        _ORACLE_V1_B2.append(OU.unet_forward(sd, V1_UNET_CFG, x2, t2, c2))
    return _ORACLE_V1_B2[0]


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "fp16x2"])
def test_v1_unet_full_size(golden, precision):
    """SEVIR-LR v1 architecture (136.8 M params), B=2 with distinct t, seeded weights; checked against the oracle run
    on this box's CPU and (sample 0 equivalent) against the reference output captured at B=1."""
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    net = CuboidTransformerUNet(**V1_UNET_CFG, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    x = seeded_input("v1x", (1, 6, 16, 16, 64), 2)
    cond = seeded_input("v1c", (1, 7, 16, 16, 64), 3)
    t = torch.tensor([500])
    out = net(x.cuda(), t.cuda(), cond.cuda())
    g = golden("v1_unet")
    e_gold = rel_l2(out, g["out_full_f16"].astype(np.float32))
    e_slice = rel_l2(out[0, :, ::4, ::4, ::8], g["out_slice"])
    print(f"[v1 {precision}] rel-L2 vs reference golden: fp16-stored full {e_gold:.3e}, fp32 slice {e_slice:.3e}")
    assert e_slice < TOL[precision]
    # batch of 2 with different timesteps against the oracle
    x2 = torch.cat([x, seeded_input("v1x2", (1, 6, 16, 16, 64), 4)])
    c2 = torch.cat([cond, seeded_input("v1c2", (1, 7, 16, 16, 64), 5)])
    t2 = torch.tensor([500, 3])
    out2 = net(x2.cuda(), t2.cuda(), c2.cuda())
    ref2 = _oracle_v1_b2(sd, x2, t2, c2)
    e = rel_l2(out2, ref2)
    print(f"[v1 {precision}] B=2 rel-L2 vs oracle {e:.3e}")
    assert e < TOL[precision]
    # sample 0 must not depend on what else is in the batch: bitwise, as long as the launches have the same K-slicing (the fp32
    # engine never splits; the bf16 engine does below 17 trajectories per launch -- csrc/igemm256.hip -- which changes the fp32
    # summation order with the batch size: then equal to fp32 round-off amplified by a few bf16 rounding boundaries)
    net.split_k = False
    assert torch.equal(net(x2.cuda(), t2.cuda(), c2.cuda())[0], net(x.cuda(), t.cuda(), cond.cuda())[0])
    net.split_k = True
    e_inv = rel_l2(out2[0], out[0])
    print(f"[v1 {precision}] sample 0 alone vs in a batch of 2 (different K-slicing in bf16 mode): rel-L2 {e_inv:.3e}")
    # a different summation order perturbs at fp32 round-off; downstream bf16 roundings amplify that to (at most) the bf16 noise level
    assert e_inv < (1e-6 if precision == "fp32" else TOL[precision] / 2)


_HEAVY = {}


# measured on MI355X (profiles/r06_*_parity_report.jsonl), rel-L2 per forward, Gaussian -> heavy-tailed weights: the bound is 2x the measured
# heavy-tailed figure.  Every engine loses the same factor (~3.5x: bf16 7.5e-3 -> 2.9e-2, fp16 1.05e-3 -> 3.3e-3), i.e. the loss is the
# conditioning of the heavy-tailed network (larger cancellations in its dot products), not an overflow / saturation of a 16-bit packer.
# fp16x2: on these weights exact (folded) weights do not help -- the folded engine measures 3.8 ... 5.1e-3 where the fp16 engine measures 3.3e-3 and the
# same engine with W_lo zeroed 3.2e-3, although every folded GEMM is exact on this distribution (kernel tests) and the Gaussian case behaves as
# predicted; DESIGN.md section 5 has the measurements (profiles/r06_h_debug_heavy_tailed.log, r06_j_wlo_effect.log).  Bounded at 2x the measured figure.
HEAVY_BOUND = {"fp32": 2e-4, "fp16x2": 1e-2, "bf16": 6e-2, "fp16": 7e-3, "fp8_conv": 0.3}


@pytest.mark.parametrize("precision,B", [("fp32", 2), ("fp16x2", 2), ("fp16x2", 32), ("bf16", 2), ("bf16", 32), ("fp16", 2), ("fp16", 32), ("fp8_conv", 2)])
def test_v1_unet_heavy_tailed_weights(precision, B):
    """Robustness of the 16-bit / 8-bit engines on checkpoint-like weights (no trained checkpoint exists offline and
    every other parity case uses Gaussian fan-in-scaled weights): Student-t(3) matrices / filters with two 30x outlier output channels
    each and two 30x entries in every norm scale (prediff_amd.seeding.heavy_tailed_state_dict), v1 size, against the oracle on the same
    weights.  B = 2 runs the small-grid forms (split-K Conv3d, 64-row / split pair kernels), B = 32 the full-occupancy ones (256-row
    tiles, the eight-wave pair form).  Bars: finite (the fp16 pair packer, common.h cvt_op4, does not saturate; the e4m3 GroupNorm
    output takes its scale from the layer's gains since round 6), the absolute bound of HEAVY_BOUND, and -- the statement that the
    16-bit / 8-bit engines are no more fragile than the arithmetic they approximate -- a Gaussian -> heavy-tailed loss factor within 2x
    of the fp32-class engine's own (same network, same inputs, 16-bit-pair operands)."""
    import json
    import os
    key = "sd"
    if key not in _HEAVY:
        tmpl = TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json")
        _HEAVY[key] = heavy_tailed_state_dict(tmpl, 77)
        _HEAVY["gauss"] = seeded_state_dict(tmpl, 1234)
        x2 = torch.cat([seeded_input("v1x", (1, 6, 16, 16, 64), 2), seeded_input("v1x2", (1, 6, 16, 16, 64), 4)])
        c2 = torch.cat([seeded_input("v1c", (1, 7, 16, 16, 64), 3), seeded_input("v1c2", (1, 7, 16, 16, 64), 5)])
        t2 = torch.tensor([500, 3])
        _HEAVY["in"] = (x2, t2, c2)
        _HEAVY["ref"] = OU.unet_forward(_HEAVY[key], V1_UNET_CFG, x2, t2, c2)
        _HEAVY["ref_gauss"] = _oracle_v1_b2(_HEAVY["gauss"], x2, t2, c2)
        assert bool(torch.isfinite(_HEAVY["ref"]).all())
    x2, t2, c2 = _HEAVY["in"]
    rep = B // 2
    xb, tb, cb = (v.repeat((rep,) + (1,) * (v.dim() - 1)).cuda() for v in (x2, t2, c2))
    errs = {}
    for kind, sd, ref in (("gauss", _HEAVY["gauss"], _HEAVY["ref_gauss"]), ("heavy", _HEAVY["sd"], _HEAVY["ref"])):
        net = CuboidTransformerUNet(**V1_UNET_CFG, precision=precision)
        net.load_state_dict(sd, strict=True)
        out = net.cuda()(xb, tb, cb)
        assert bool(torch.isfinite(out).all()), f"{precision} B={B} {kind}: non-finite output"
        errs[kind] = rel_l2(out[:2], ref)
        errs[kind + "_max"] = max(rel_l2(out[2 * r:2 * r + 2], ref) for r in range(rep))
        del net
    print(f"[v1 heavy-tailed {precision} B={B}] rel-L2 vs oracle: Gaussian weights {errs['gauss']:.3e}, Student-t(3) + 30x outlier channels "
          f"{errs['heavy']:.3e} (worst pair of the batch {errs['heavy_max']:.3e})")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps(dict(test="v1_unet_heavy_tailed", precision=precision, B=B, **errs)) + "\n")
    assert errs["heavy_max"] < HEAVY_BOUND[precision]
    _HEAVY[("ratio", precision, B)] = errs["heavy_max"] / errs["gauss"]
    r32 = _HEAVY.get(("ratio", "fp32", 2))          # (the fp32-class engine's loss factor, measured at 2 trajectories: it does not depend on the batch)
    if r32 is not None and precision != "fp32":
        print(f"[v1 heavy-tailed {precision} B={B}] loss factor {errs['heavy_max'] / errs['gauss']:.2f} (fp32-class engine: {r32:.2f})")
        # (e4m3: 3 mantissa bits meet the outlier channels' dynamic range -- reported, bounded above;  fp16x2: its Gaussian figure has no
        #  weight term while its heavy-tailed figure is the fp16 engine's -- the RATIO says nothing about robustness there, the bound does)
        if precision not in ("fp8_conv", "fp16x2"):
            assert errs["heavy_max"] / errs["gauss"] < 2.0 * max(r32, 1.0)


@pytest.mark.parametrize("name", ["axial", "v1"])
def test_folded_engine_never_runs_one_product_kernels(name):
    """precision="fp16x2": the round-3 fused token kernels (pd_attn_block_fused, pd_ffn_fused) stream ONE weight image -- handed a folded
    operand they would silently read W_hi only.  The engine must refuse them whatever the A/B flags say (bench.py once re-enabled them and
    measured a faster, one-product engine): a forward with the flags forced on equals the forward with them off bit for bit, and differs
    from the one-product fp16 engine."""
    cfg = V1_UNET_CFG if name == "v1" else TINY_UNET_CFGS[name]
    sd = seeded_state_dict(TP.unet_template(cfg, "v1_unet_schema.json" if name == "v1" else "tiny_unet_schema.json", None if name == "v1" else name), 31)
    x = seeded_input("fx", (2,) + tuple(cfg["target_shape"]), 2).cuda()
    cond = seeded_input("fc", (2,) + tuple(cfg["input_shape"]), 3).cuda()
    t = torch.tensor([7, 431]).cuda()
    outs = {}
    for key, prec, flags in (("off", "fp16x2", False), ("on", "fp16x2", True), ("fp16", "fp16", None)):
        net = CuboidTransformerUNet(**cfg, precision=prec)
        net.load_state_dict(sd, strict=True)
        if flags is not None:
            net.fuse_pair = False                    # the LayerNorm / folded pd_igemm / attention-core launches
            net.fuse_attn = net.fuse_ffn = flags
        outs[key] = net.cuda()(x, t, cond)
        del net
    assert torch.equal(outs["on"], outs["off"])
    assert rel_l2(outs["off"], outs["fp16"]) > 1e-5          # (the folded engine is not the one-product engine)


def test_repack_after_weight_update():
    cfg = TINY_UNET_CFGS["axial"]
    net = CuboidTransformerUNet(**cfg).cuda()
    x = seeded_input("rx", (1,) + tuple(cfg["target_shape"]), 2).cuda()
    c = seeded_input("rc", (1,) + tuple(cfg["input_shape"]), 3).cuda()
    t = torch.tensor([5]).cuda()
    assert float(net(x, t, c).abs().max()) == 0          # default init -> exactly zero output
    sd = seeded_state_dict(net.state_dict(), 77)
    net.load_state_dict(sd)
    out = net(x, t, c)
    assert float(out.abs().max()) > 0
    ref = OU.unet_forward({k: v.cpu() for k, v in sd.items()}, cfg, x.cpu(), t.cpu(), c.cpu())
    assert rel_l2(out, ref) < TOL["bf16"]


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_v1_forward_repeats_bit_equal_at_full_occupancy(precision):
    """32 trajectories through the v1 denoiser, ten times: the fused level-0 kernels run 1664 workgroups, two resident per CU, the
    Conv3d / GEMM kernels several rounds of tiles.  Every repeat must be bit-equal to the first: a timing-dependent fault (the kind
    profiles/r03_h_conv2d_gn_hazard.md records for an early build of the VAE kernel: wrong 16-lane groups in random workgroups, only
    with co-resident workgroups) would show here even where a rel-L2 bound against the oracle would not."""
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    net = CuboidTransformerUNet(**V1_UNET_CFG, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    g = torch.Generator(device="cpu").manual_seed(5)
    B = 32
    x = torch.randn((B, 6, 16, 16, 64), generator=g).cuda()
    cond = torch.randn((B, 7, 16, 16, 64), generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    first = net(x, t, cond).clone()
    assert bool(torch.isfinite(first).all())
    for k in range(9):
        out = net(x, t, cond)
        assert torch.equal(out, first), f"repeat {k + 1} differs: max |d| {float((out - first).abs().max()):.3e}"


def test_v1_forward_b32_vs_oracle_with_pair_kernel():
    """The occupancy the benchmark runs at -- 32 trajectories per launch, bf16 engine, every (attention, FFN) pair on
    pd_attn_ffn_pair (csrc/pair_block.hip: level 0 832 tiles of 128 rows, four per workgroup; level 1 416 / 512 tiles of 64 rows) -- against the oracle's CPU forward of the
    same 32 samples with 32 different timesteps (not a self-comparison), and against the engine with the pair kernel switched off
    (the two round-3 kernels per pair)."""
    from prediff_amd import _lib as L
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    net = CuboidTransformerUNet(**V1_UNET_CFG, precision="bf16")
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    B = 32
    x = seeded_input("b32x", (B, 6, 16, 16, 64), 2)
    cond = seeded_input("b32c", (B, 7, 16, 16, 64), 3)
    t = (torch.arange(B) * 31 + 5) % 1000
    calls = []
    real = L.attn_ffn_pair
    L.attn_ffn_pair = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        assert net.fuse_pair and (B * 256 + 7) // 8 >= net.pair_min_tiles
        out = net(x.cuda(), t.cuda(), cond.cuda())
        n_pair = len(calls)
        net.fuse_pair = False
        out_r3 = net(x.cuda(), t.cuda(), cond.cuda())
        assert len(calls) == n_pair
    finally:
        L.attn_ffn_pair = real
    assert n_pair == 48, f"{n_pair} pair launches (expected the 24 level-0 + 24 level-1 pairs of depth [4, 4] x down / up x 3 axes)"
    # the oracle's CPU forward on 12 of the 32 samples (first / last of the launch and one from every tile group: a sample's rows do not
    # depend on its neighbours, so its oracle forward at B = 12 is its oracle forward at B = 32) -- 25 s instead of 70 s of CPU time
    idx = torch.tensor([0, 1, 2, 5, 9, 13, 16, 20, 23, 27, 30, 31])
    ref = OU.unet_forward(sd, V1_UNET_CFG, x[idx], t[idx], cond[idx])
    o, o3 = out.cpu()[idx], out_r3.cpu()[idx]
    e, e3 = rel_l2(o, ref), rel_l2(o3, ref)
    e_ab = rel_l2(out, out_r3.cpu())
    print(f"[v1 bf16 B=32] rel-L2 vs oracle (12 of 32 samples): pair kernel {e:.3e}, round-3 kernels {e3:.3e}; between the two (all 32) {e_ab:.3e}")
    assert e < TOL["bf16"] and e3 < TOL["bf16"]
    assert e_ab < TOL["bf16"]
    per_sample = [rel_l2(o[i], ref[i]) for i in range(len(idx))]
    assert max(per_sample) < 2 * TOL["bf16"], per_sample
    # every sample against the engine without the pair kernel (all 32: catches a wrong tile anywhere in the launch)
    per_ab = [rel_l2(out[i], out_r3[i].cpu()) for i in range(B)]
    assert max(per_ab) < 2 * TOL["bf16"], per_ab


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["5x7x6", "22x5x6"])
def test_tiny_unet_nearest_padding_non_divisible(precision, case):
    """padding_type="nearest" on a grid neither the cuboids (2, 4, 4) nor the (1, 2, 2) patch merging divide (5 x 7 x 6): the gather /
    receive token tables of the attention layers and the nearest-padded patch merging, whole denoiser against the oracle
    (reference models/utils.py:228-270, cuboid_transformer.py:261-296, :812-966).  Case 22x5x6: temporal cuboids of 13 on 22 frames
    (padded to 26) and 5 rows merged by 2 (padded to 6) -- sizes where torch's floor(dst * float32(in / out)) and the integer
    floor(dst * in / out) pick different tokens."""
    from _cases import _unet
    if case == "5x7x6":
        cfg = _unet("video_swin_2x4", padding_type="nearest", input_shape=[3, 7, 6, 4], target_shape=[2, 7, 6, 4])
    else:
        cfg = _unet(None, padding_type="nearest", input_shape=[15, 5, 6, 4], target_shape=[7, 5, 6, 4],
                    block_cuboid_size=[(13, 2, 2), (13, 2, 2)], block_cuboid_strategy=[("l", "l", "l"), ("l", "l", "l")],
                    block_cuboid_shift_size=[(0, 0, 0), (0, 0, 0)])
    net = CuboidTransformerUNet(**cfg, precision=precision)
    sd = seeded_state_dict(net.state_dict(), 911)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    x = seeded_input("npx", (2,) + tuple(cfg["target_shape"]), 2)
    cond = seeded_input("npc", (2,) + tuple(cfg["input_shape"]), 3)
    t = torch.tensor([7, 431])
    out = net(x.cuda(), t.cuda(), cond.cuda())
    ref = OU.unet_forward({k: v.cpu() for k, v in sd.items()}, cfg, x, t, cond)
    e = rel_l2(out, ref)
    print(f"[tiny unet, nearest padding, {case}, {precision}] rel-L2 vs oracle {e:.3e}")
    assert e < TOL[precision]
