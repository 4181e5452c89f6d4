"""GPU: the BASELINE.json configurations end to end at their real sizes ("configs untested").

config 1  N-body stand-in, 1 sample, DDIM-10                         -> test_nbody_standin
config 2  SEVIR-LR v1, DDIM-50: fp32 engine vs the oracle loop on the same noise tape (north_star bar 1e-3), bf16 drift reported;
          all-zero and 90 %-sparse contexts                           -> test_v1_ddim50_vs_oracle, test_degenerate_contexts
config 3  ensemble sharding at v1 size incl. VAE and the RCCL all-gather (world of one) -> test_v1_ensemble_rccl_world1
config 4  knowledge-aligned ancestral step at v1 size, t in {99, 0}, fp32 and bf16 vs the reference golden -> test_v1_aligned_step
config 5  full-resolution geometry (latent 25 x 48 x 48, axial cuboids 25 / 48 / 48 and 25 / 24 / 24): one denoiser forward vs the
          oracle, fp32 and bf16 operands (the fp8 operand path of that config is not built)         -> test_fullres_forward
"""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _templates as TP  # noqa: E402
from _cases import (FULLRES_UNET_CFG, NBODY_LDM_KW, NBODY_UNET_CFG, NBODY_VAE_CFG, V1_ALIGN_ARGS, V1_LDM_KW, V1_UNET_CFG,  # noqa: E402
                    V1_VAE_CFG)
from _weights import seeded_input, seeded_state_dict  # noqa: E402
from oracle import diffusion as OD  # noqa: E402
from oracle import unet as OU  # noqa: E402
from oracle import vae as OV  # noqa: E402
from prediff_amd.autoencoder_kl import AutoencoderKL  # noqa: E402
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet  # noqa: E402
from prediff_amd.latent_diffusion import LatentDiffusion  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _report(name, **vals):
    """Measured parity numbers for DESIGN.md: appended to gpurun_out/parity_report.jsonl (scratch; merged back by gpurun)."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **vals)) + "\n")


_V1_SD = {}


def _v1_unet_sd():
    if "u" not in _V1_SD:
        _V1_SD["u"] = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    return _V1_SD["u"]


def _v1_ldm(precision, vae=False):
    net = CuboidTransformerUNet(**V1_UNET_CFG, precision=precision)
    net.load_state_dict(_v1_unet_sd(), strict=True)
    v = None
    if vae:
        v = AutoencoderKL(**V1_VAE_CFG, precision=precision)
        v.load_state_dict(seeded_state_dict(TP.from_schema("v1_vae_schema.json"), 4321), strict=True)
    ldm = LatentDiffusion(torch_nn_module=net, first_stage_model=v, cond_stage_model=("__is_first_stage__" if vae else None), **V1_LDM_KW)
    return ldm.cuda().eval()


# ------------------------------------------------------------------------------------------------ config 1
@pytest.mark.parametrize("precision", ["fp32", "fp16x2", "fp16", "bf16"])
def test_nbody_standin(golden, precision):
    g = golden("nbody")
    net = CuboidTransformerUNet(**NBODY_UNET_CFG, precision=precision)
    net.load_state_dict(seeded_state_dict(TP.unet_template(NBODY_UNET_CFG, "nbody_schema.json", "unet"), 800), strict=True)
    vae = AutoencoderKL(**NBODY_VAE_CFG, precision=precision)
    vae.load_state_dict(seeded_state_dict(TP.from_schema("nbody_schema.json", "vae"), 801), strict=True)
    ldm = LatentDiffusion(torch_nn_module=net, first_stage_model=vae, cond_stage_model="__is_first_stage__", **NBODY_LDM_KW).cuda().eval()
    y = seeded_input("nby", (1, 10, 64, 64, 1), 0, kind="uniform").cuda()
    xT = seeded_input("nbxT", (1, 10, 16, 16, 4), 1).cuda()
    zc = ldm.cond_stage_forward({"y": y})
    lat = ldm.sample(cond={"y": y}, batch_size=1, sampler="ddim", ddim_steps=10, eta=0.0, x_T=xT, return_decoded=False)
    dec = ldm.sample(cond={"y": y}, batch_size=1, sampler="ddim", ddim_steps=10, eta=0.0, x_T=xT)
    e = dict(zc=rel_l2(zc, g["zc"]), latent=rel_l2(lat, g["latent"]), decoded=rel_l2(dec, g["decoded"]))
    print(f"[nbody {precision}] rel-L2 vs reference modules: {e}")
    _report("nbody_standin", precision=precision, **e)
    assert dec.shape == (1, 10, 64, 64, 1)
    tol = {"fp32": 1e-3, "fp16x2": 4e-3, "fp16": 6e-3, "bf16": 5e-2}[precision]          # (measured bf16: 9e-3; fp16 is asked to be 8x finer with margin;
    # fp16x2: the denoiser's weights exact, the VAE -- which runs once -- on the one-product fp16 engine)
    assert max(e.values()) < tol


# ------------------------------------------------------------------------------------------------ config 2
def test_v1_ddim50_vs_oracle():
    """v1 size, DDIM-50 (eta 0), one trajectory, one noise tape: the fp32-class engine against the oracle loop run on this box's CPU within the
    north_star bar (1e-3 rel-L2 after all 50 steps); the bf16 throughput mode's drift over the same horizon is measured and
    reported (the reference is fp32 only: SURVEY.md F7 -- no 1e-3 claim for bf16).  The DDIM update rule itself is parity-unpinned
    (no reference implementation, F3); the denoiser inside it is pinned."""
    B = 1
    sd = _v1_unet_sd()
    zc = seeded_input("d50c", (B, 7, 16, 16, 64), 21)
    xT = seeded_input("d50x", (B, 6, 16, 16, 64), 22)
    tape = [xT] + [torch.zeros_like(xT)] * 50
    ac = np.cumprod(1.0 - OD.beta_schedule("linear", 1000)).astype(np.float32)
    t0 = time.time()
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 32))          # many-core hosts oversubscribe on these small convolutions (bench.py:cpu_baseline)
    try:
        with torch.no_grad():
            traj = OD.ddim_sample_loop(ac, lambda z, t, c: OU.unet_forward(sd, V1_UNET_CFG, z, t, c), zc, tape, 50, eta=0.0)
    finally:
        torch.set_num_threads(nthr)
    t_cpu = time.time() - t0
    ref = traj[-1]
    errs = {}
    for precision in ("fp32", "fp16x2", "fp16x2_lin", "fp16", "bf16", "fp8_conv", "fp8"):
        ldm = _v1_ldm(precision)
        out, inter = ldm.sample(cond=zc.cuda(), batch_size=B, sampler="ddim", ddim_steps=50, eta=0.0, x_T=xT.cuda(),
                                return_decoded=False, return_intermediates=True)
        assert bool(torch.isfinite(out).all())          # (fp16 operands saturate at 65504: a finite result is part of the check)
        errs[precision] = rel_l2(out, ref)
        errs[precision + "_by_step"] = [round(rel_l2(inter[k], traj[k]), 6) for k in (1, 10, 25, 40, 50)]
        # the graph/lane path used by the benchmark gives the same trajectory
        out2 = ldm.sample(cond=zc.cuda(), batch_size=B, sampler="ddim", ddim_steps=50, eta=0.0, x_T=xT.cuda(), return_decoded=False)
        assert torch.equal(out2, out)
        del ldm
    print(f"[v1 DDIM-50] rel-L2 vs oracle loop after 50 steps: fp32 {errs['fp32']:.3e}, fp16x2 (fp16 activations x hi+lo fp16 weights) {errs['fp16x2']:.3e} "
          f"by step {errs['fp16x2_by_step']}, fp16x2_lin (Conv3d on one product) {errs['fp16x2_lin']:.3e} by step {errs['fp16x2_lin_by_step']}, "
          f"fp16 (IEEE-half operands: the TF32 class) {errs['fp16']:.3e} "
          f"by step {errs['fp16_by_step']}, bf16 {errs['bf16']:.3e}, fp8_conv (e4m3 Conv3d) "
          f"{errs['fp8_conv']:.3e}, fp8 (e4m3 Conv3d + K >= 512 linears) {errs['fp8']:.3e}; by step (1,10,25,40,50): fp32 {errs['fp32_by_step']} "
          f"bf16 {errs['bf16_by_step']} fp8_conv {errs['fp8_conv_by_step']} fp8 {errs['fp8_by_step']}; oracle loop {t_cpu:.0f} s on CPU")
    _report("v1_ddim50", oracle_cpu_s=round(t_cpu, 1), **errs)
    assert errs["fp32"] < 1e-3
    # IEEE-half activations x (hi + lo) IEEE-half weights, two MFMA products: the north-star bar with a single-pass activation path
    # (measured 4e-4: the fp16 engine's 1.3e-3 minus its weight-rounding term, test_v1_fp16_error_budget)
    assert errs["fp16x2"] < 1e-3 and errs["fp16x2"] < 0.5 * errs["fp16"]
    # ... and with the 3x3x3 convolutions left on one product (their weight rounding is 3.1e-4 of the 8.8e-4 weight term of a forward)
    assert errs["fp16x2_lin"] < 1e-3 and errs["fp16x2"] <= errs["fp16x2_lin"] < errs["fp16"]
    # IEEE-half operands: 8x finer significands than bf16 (expected ~1.2e-3 where bf16 measures 1.0e-2); the bar below is 2x that
    assert errs["fp16"] < 2.5e-3 and errs["fp16"] < 0.3 * errs["bf16"]
    # guard rails at 2x what is measured (bf16 1.0e-2, fp8 3.8e-2 after all 50 steps: DESIGN.md §4), so that a regression shows
    assert errs["bf16"] < 2e-2 and np.isfinite(errs["bf16"])
    assert errs["fp8_conv"] < 8e-2 and np.isfinite(errs["fp8_conv"])          # report-only operand types (BASELINE config 5): measured
    assert errs["fp8"] < 0.16 and np.isfinite(errs["fp8"])                    # 3.8e-2 (convolutions) / 9.2e-2 (+ the level-1 linears)


def test_v1_fp16_error_budget():
    """Where the fp16 engine's DDIM-50 error (1.3e-3: just over the 1e-3 bar) comes from -- the measurement behind precision="fp16x2"
    ("a <= 1e-3 engine that is not 3x slower").  The oracle loop is run a second time with every matrix / filter
    rounded to IEEE half exactly as the engine's weight packs round them: against THAT loop the engine's deviation contains no
    weight-rounding term -- it is what an engine with exact weights measures against the true oracle -- and the two oracle loops
    against each other give the weight term alone (fp32 arithmetic on rounded weights)."""
    B = 1
    sd = _v1_unet_sd()
    sd16 = {k: (v.half().float() if (torch.is_floating_point(v) and v.dim() >= 2) else v) for k, v in sd.items()}
    zc = seeded_input("d50c", (B, 7, 16, 16, 64), 21)
    xT = seeded_input("d50x", (B, 6, 16, 16, 64), 22)
    tape = [xT] + [torch.zeros_like(xT)] * 50
    ac = np.cumprod(1.0 - OD.beta_schedule("linear", 1000)).astype(np.float32)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 32))
    try:
        with torch.no_grad():
            ref = OD.ddim_sample_loop(ac, lambda z, t, c: OU.unet_forward(sd, V1_UNET_CFG, z, t, c), zc, tape, 50, eta=0.0)
            ref16 = OD.ddim_sample_loop(ac, lambda z, t, c: OU.unet_forward(sd16, V1_UNET_CFG, z, t, c), zc, tape, 50, eta=0.0)
    finally:
        torch.set_num_threads(nthr)
    ldm = _v1_ldm("fp16")
    out, inter = ldm.sample(cond=zc.cuda(), batch_size=B, sampler="ddim", ddim_steps=50, eta=0.0, x_T=xT.cuda(), return_decoded=False,
                            return_intermediates=True)
    steps = (1, 10, 25, 40, 50)
    e_true = [rel_l2(inter[k], ref[k]) for k in steps]             # activation + weight rounding
    e_w16 = [rel_l2(inter[k], ref16[k]) for k in steps]            # activation rounding only (the oracle shares the rounded weights)
    e_wonly = [rel_l2(ref16[k], ref[k]) for k in steps]            # weight rounding only (fp32 arithmetic on rounded weights)
    print(f"[v1 DDIM-50 fp16 budget] by step {steps}: engine vs oracle {[f'{e:.2e}' for e in e_true]}; engine vs oracle-on-fp16-weights "
          f"(activation term) {[f'{e:.2e}' for e in e_w16]}; oracle-on-fp16-weights vs oracle (weight term) {[f'{e:.2e}' for e in e_wonly]}")
    _report("v1_fp16_error_budget", steps=list(steps), engine_vs_oracle=e_true, activation_term=e_w16, weight_term=e_wonly)
    assert e_true[-1] < 2.5e-3
    # measured (round 6): 1.31e-3 = 3.9e-4 (activations: fresh rounding noise at every step) (+) 1.28e-3 (weights: ONE fixed perturbation of the
    # network, the same bias at each of the 50 steps) -- the weight term dominates and grows with the step count, the activation term does
    # not.  Hence precision="fp16x2": exact (hi + lo) weights, once-rounded activations, two products.
    assert e_w16[-1] < 0.5 * e_true[-1] and e_w16[-1] < 1e-3 and e_wonly[-1] > e_w16[-1]


@pytest.mark.parametrize("kind", ["zeros", "sparse90"])
def test_degenerate_contexts(kind):
    """Real VIL frames are uint8/255 with many zeros: an all-zero and a 90 %-sparse context through the VAE
    encoder (GroupNorm on constant input: fp64 statistics, eps) and one denoiser forward, against the oracle."""
    vsd = seeded_state_dict(TP.from_schema("v1_vae_schema.json"), 4321)
    y = torch.zeros(1, 7, 128, 128, 1)
    if kind == "sparse90":
        g = torch.Generator().manual_seed(31)
        y = (torch.randint(0, 256, y.shape, generator=g).float() / 255) * (torch.rand(y.shape, generator=g) > 0.9)
    frames = y.permute(0, 1, 4, 2, 3).reshape(7, 1, 128, 128)
    with torch.no_grad():
        zc_ref = OV.vae_encode_mode(vsd, V1_VAE_CFG, frames).reshape(1, 7, 64, 16, 16).permute(0, 1, 3, 4, 2).contiguous()
    x = seeded_input("dgx", (1, 6, 16, 16, 64), 23)
    t = torch.tensor([981])
    with torch.no_grad():
        eps_ref = OU.unet_forward(_v1_unet_sd(), V1_UNET_CFG, x, t, zc_ref)
    for precision, tol_z, tol_e in (("fp32", 1e-4, 2e-4), ("bf16", 2e-2, 3e-2)):
        ldm = _v1_ldm(precision, vae=True)
        zc = ldm.cond_stage_forward({"y": y.cuda()})
        eps = ldm.apply_model(x.cuda(), t.cuda(), zc)
        ez, ee = rel_l2(zc, zc_ref), rel_l2(eps, eps_ref)
        print(f"[{kind} {precision}] context latent {ez:.3e}, eps {ee:.3e} (vs oracle)")
        _report("degenerate_context", kind=kind, precision=precision, zc=ez, eps=ee)
        assert bool(torch.isfinite(zc).all()) and ez < tol_z and ee < tol_e
        del ldm


# ------------------------------------------------------------------------------------------------ config 3
def test_v1_ensemble_rccl_world1():
    """sample_ensemble at v1 size incl. the VAE, 4 members on one GPU, inside a world-of-one "nccl" (= RCCL) process group so that
    all_gather_into_tensor really runs; members must equal the same members sampled without a process group."""
    import torch.distributed as dist
    from prediff_amd.ensemble import sample_ensemble
    ldm = _v1_ldm("bf16", vae=True)
    y = seeded_input("ensy", (1, 7, 128, 128, 1), 24, kind="uniform").cuda()
    kw = dict(base_seed=1000, sampler="ddim", ddim_steps=4, eta=1.0)
    plain = sample_ensemble(ldm, {"y": y}, 4, **kw)
    assert plain.shape == (4, 6, 128, 128, 1) and bool(torch.isfinite(plain).all())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        gathered = sample_ensemble(ldm, {"y": y}, 4, force_collective=True, **kw)
        probe = torch.arange(8, dtype=torch.float32, device="cuda").reshape(4, 2)
        from prediff_amd.ensemble import all_gather_members
        assert torch.equal(all_gather_members(probe, 4, 0, 1, force_collective=True), probe)
    finally:
        dist.destroy_process_group()
    assert torch.equal(gathered, plain)
    assert rel_l2(plain[0], plain[1]) > 1e-3        # members differ


# ------------------------------------------------------------------------------------------------ config 4
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16+bf16guidance"])
def test_v1_aligned_step(golden, precision):
    """precision "bf16+bf16guidance": the bf16 denoiser with the guidance network's convolutions on single-pass bf16 operands too
    (SEVIRAvgIntensityAlignment(hip_precision="bf16")); the other two keep the guidance at fp32-class accuracy."""
    from prediff_amd.alignment import SEVIRAvgIntensityAlignment
    g = golden("v1_aligned")
    guidance_precision = "bf16" if precision.endswith("guidance") else "fp32"
    precision = precision.split("+")[0]
    ldm = _v1_ldm(precision)
    al = SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid", model_args=dict(V1_ALIGN_ARGS),
                                    hip_precision=guidance_precision)
    al.model.load_state_dict(seeded_state_dict(al.model.state_dict(), 701))
    al.model.cuda()
    ldm.set_alignment(al.get_mean_shift)
    B = 2
    zt, zc = seeded_input("v1azt", (B, 6, 16, 16, 64), 12).cuda(), seeded_input("v1azc", (B, 7, 16, 16, 64), 13).cuda()
    avg = torch.as_tensor(g["avg_x_gt"]).cuda()
    tol = 2e-4 if precision == "fp32" else 2e-2
    for tt in (99, 0):
        t = torch.full((B,), tt, dtype=torch.long, device="cuda")
        noise = seeded_input(f"v1an{tt}", (B, 6, 16, 16, 64), 14).cuda()
        out = ldm.p_sample(zt=zt, zc=zc, t=t, y=None, use_alignment=True, alignment_kwargs={"avg_x_gt": avg}, noise=noise)
        e = rel_l2(out[:, :, ::2, ::2, ::4], g[f"out_{tt}_slice"])
        cs = abs(float(out.double().abs().sum()) / float(g[f"out_{tt}_abs_sum"][0]) - 1)
        print(f"[v1 aligned {precision}, guidance convolutions {guidance_precision} t={tt}] rel-L2 vs reference {e:.3e}, |.|-sum deviation {cs:.2e}")
        _report("v1_aligned_step", precision=precision, guidance=guidance_precision, t=tt, rel_l2=e)
        assert e < tol and cs < tol
    # the looped form (sample(timesteps=2, use_alignment=True): denoiser graphs on the lane streams overlapping the autograd guidance)
    tape = [zt] + [seeded_input(f"v1an{tt}", (B, 6, 16, 16, 64), 14).cuda() for tt in (99, 0)]
    ldm.aligned_lanes = 2
    a = ldm.sample(cond=zc, batch_size=B, timesteps=2, use_alignment=True, alignment_kwargs={"avg_x_gt": avg}, return_decoded=False,
                   noise_tape=tape)
    ldm.use_hip_graph = False
    b = ldm.sample(cond=zc, batch_size=B, timesteps=2, use_alignment=True, alignment_kwargs={"avg_x_gt": avg}, return_decoded=False,
                   noise_tape=tape)
    if precision == "fp32":
        # same arithmetic on both paths (30 fresh captures of scripts/stress_aligned_determinism.py: bit-identical every time); one run
        # in ~40 on the build boxes differed in the last bits, so a 1e-6 band is accepted here and reported
        if not torch.equal(a, b):
            e = rel_l2(a, b)
            print(f"[v1 aligned fp32] lanes vs eager not bit-identical in this run: rel-L2 {e:.3e}, max |diff| {float((a - b).abs().max()):.3e}")
            assert e < 1e-6
    else:
        # bf16 engine, fewer than 17 trajectories per launch: the lanes (1 trajectory each here) and the eager batch of 2 use
        # different K-slicings of the split-K Conv3d (csrc/igemm256.hip) -> fp32 summation order differs -> equal to bf16 noise
        e = rel_l2(a, b)
        print(f"[v1 aligned bf16] lanes of 1 vs eager batch of 2: rel-L2 {e:.3e}")
        assert e < 1e-2


# ------------------------------------------------------------------------------------------------ config 5 (geometry; bf16 / fp32 operands)
def test_fullres_forward():
    """The v1 denoiser on the full-resolution latent grid (13 + 12 frames of 48 x 48 x 64: SURVEY.md §8(d) row 5): cuboid volumes
    25 and 48 at level 0 (beyond the fused block kernel's 16: LayerNorm -> QKV GEMM -> generic attention core -> proj GEMM),
    25 / 24 / 24 at level 1 (head_dim 128).  One forward of one trajectory against the oracle (11.4 TFLOP on the CPU)."""
    # checkpoint schema of this geometry from the module itself (position tables / relative-position tables and index buffers grow
    # with the grid; the index buffers are the ones pinned against the reference's at the v1 and tiny shapes)
    sd = seeded_state_dict(CuboidTransformerUNet(**FULLRES_UNET_CFG).state_dict(), 1234)
    x = seeded_input("frx", (1, 12, 48, 48, 64), 41)
    zc = seeded_input("frc", (1, 13, 48, 48, 64), 42)
    t = torch.tensor([500])
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 64))
    t0 = time.time()
    try:
        with torch.no_grad():
            ref = OU.unet_forward(sd, FULLRES_UNET_CFG, x, t, zc)
    finally:
        torch.set_num_threads(nthr)
    t_cpu = time.time() - t0
    # "fp8_conv" / "fp8": e4m3 operands (3 mantissa bits: report-only accuracy, BASELINE config 5) for the convolutions / also for the
    # K >= 512 linears; bounds at 2x what is measured (bf16 7.4e-3, fp8_conv 4.3e-2)
    # "fp16x2": the folded-weight engine on this geometry (no pair kernel: LayerNorm / folded pd_igemm / multi-key-tile attention core launches)
    for precision, tol in (("fp32", 1e-4), ("fp16x2", 1.5e-3), ("bf16", 1.5e-2), ("fp8_conv", 8e-2), ("fp8", 0.14)):
        net = CuboidTransformerUNet(**FULLRES_UNET_CFG, precision=precision)
        net.load_state_dict(sd, strict=True)
        net = net.cuda()
        out = net(x.cuda(), t.cuda(), zc.cuda())
        e = rel_l2(out, ref)
        print(f"[fullres {precision}] one forward rel-L2 vs oracle {e:.3e} (oracle {t_cpu:.0f} s on CPU)")
        _report("fullres_forward", precision=precision, rel_l2=e, oracle_cpu_s=round(t_cpu, 1))
        assert out.shape == (1, 12, 48, 48, 64) and e < tol
        del net


def test_fullres_ddim_chain():
    """Config 5 over a horizon, not one forward: a short DDIM chain (ddim_steps = 3 of the reference's uniform schedule helper = the four
    timesteps 999, 667, 334, 1 of the 1000-step schedule, eta 0) of one trajectory on the full-resolution latent grid through
    LatentDiffusion.sample, against the oracle loop on the same x_T (4 oracle forwards of 11.4 TFLOP on this box's CPU).  fp32-class engine within the north-star bar; bf16 and the two e4m3 modes bounded at 2x what is
    measured."""
    from prediff_amd.presets import FULLRES_LDM_KW
    sd = seeded_state_dict(CuboidTransformerUNet(**FULLRES_UNET_CFG).state_dict(), 1234)
    zc = seeded_input("frc", (1, 13, 48, 48, 64), 42)
    xT = seeded_input("frxT", (1, 12, 48, 48, 64), 43)
    tape = [xT] + [torch.zeros_like(xT)] * 8
    ac = np.cumprod(1.0 - OD.beta_schedule("linear", 1000)).astype(np.float32)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 64))
    t0 = time.time()
    try:
        with torch.no_grad():
            traj = OD.ddim_sample_loop(ac, lambda z, t, c: OU.unet_forward(sd, FULLRES_UNET_CFG, z, t, c), zc, tape, 3, eta=0.0)
    finally:
        torch.set_num_threads(nthr)
    t_cpu = time.time() - t0
    errs = {}
    for precision, tol in (("fp32", 1e-3), ("bf16", 1.7e-2), ("fp8_conv", 9e-2), ("fp8", 0.15)):      # measured 1.3e-5 / 8.2e-3 / 4.3e-2 / 7.5e-2
        net = CuboidTransformerUNet(**FULLRES_UNET_CFG, precision=precision)
        net.load_state_dict(sd, strict=True)
        ldm = LatentDiffusion(torch_nn_module=net, first_stage_model=None, cond_stage_model=None, **FULLRES_LDM_KW).cuda().eval()
        out, inter = ldm.sample(cond=zc.cuda(), batch_size=1, sampler="ddim", ddim_steps=3, eta=0.0, x_T=xT.cuda(), return_decoded=False,
                                return_intermediates=True)
        assert len(inter) == len(traj)
        errs[precision] = [round(rel_l2(inter[k], traj[k]), 6) for k in range(1, len(traj))]
        print(f"[fullres DDIM-3 {precision}] rel-L2 vs the oracle loop after each of the {len(traj) - 1} steps: {errs[precision]} (oracle {t_cpu:.0f} s on CPU)")
        assert out.shape == (1, 12, 48, 48, 64) and errs[precision][-1] < tol and np.isfinite(errs[precision][-1])
        del ldm, net
    _report("fullres_ddim3", oracle_cpu_s=round(t_cpu, 1), **errs)


def test_v1_unet_fp8_conv(golden):
    """precision="fp8" at the v1 size: GroupNorm -> SiLU -> e4m3 rows -> scaled-MFMA Conv3d for the 34 convolutions of a forward, the
    rest of the bf16 engine unchanged.  Against the reference golden (fp32 slice) and against the bf16 engine; report-only accuracy."""
    import _templates as TP
    g = golden("v1_unet")
    sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
    x, cond, t = seeded_input("v1x", (1, 6, 16, 16, 64), 2).cuda(), seeded_input("v1c", (1, 7, 16, 16, 64), 3).cuda(), torch.tensor([500]).cuda()
    outs = {}
    for precision in ("bf16", "fp8_conv", "fp8", "fp8+linears"):
        net = CuboidTransformerUNet(**V1_UNET_CFG, precision=precision.split("+")[0])
        if precision == "fp8+linears":
            # one trajectory: the level-1 pairs of the default engine take the bf16 split form of the pair kernel (small grids), which is
            # faster AND more exact than seven launches with e4m3 linears; switched off here so that those launches are exercised
            net.pair_split = False
        net.load_state_dict(sd, strict=True)
        outs[precision] = net.cuda()(x, t, cond)
        e = rel_l2(outs[precision][0, :, ::4, ::4, ::8], g["out_slice"])
        print(f"[v1 unet {precision}] rel-L2 vs the reference golden (fp32 slice) {e:.3e}")
        _report("v1_unet_fp8_conv", precision=precision, rel_l2=e)
    assert torch.isfinite(outs["fp8"]).all() and torch.isfinite(outs["fp8_conv"]).all()
    e8c, e8, e8l = (rel_l2(outs[k], outs["bf16"]) for k in ("fp8_conv", "fp8", "fp8+linears"))
    print(f"[v1 unet fp8] rel-L2 vs the bf16 engine: e4m3 convolutions {e8c:.3e}, precision='fp8' as the engine runs one trajectory {e8:.3e}, "
          f"with the e4m3 K >= 512 linears forced {e8l:.3e}")
    assert 1e-4 < e8c < 8e-2          # different arithmetic (not the bf16 path by accident), same function (measured 3.3e-2)
    assert e8 < 0.14
    assert e8c < e8l < 0.14           # the linears add their own 3-mantissa-bit noise (measured 6e-2)


# ------------------------------------------------------------------------------------------------ config 4, the whole chain
def test_v1_aligned_chain_100():
    """BASELINE config 4 over its full horizon: 100 knowledge-aligned ancestral steps (t = 99 ... 0, guide_scale 50) at the v1 size,
    B = 2, one noise tape.  Oracle = oracle.diffusion.ddpm_sample_loop around the oracle denoiser and the guidance network's PyTorch
    CPU path (the form pinned against the reference golden by tests/test_alignment.py).  The fp32-class engine has to stay inside the
    north_star bar (1e-3 rel-L2) over all 100 steps; the bf16 engine's drift over the same chain is measured and bounded."""
    from prediff_amd.alignment import SEVIRAvgIntensityAlignment
    B, T = 2, 100
    sd = _v1_unet_sd()
    zc = seeded_input("c4c", (B, 7, 16, 16, 64), 51)
    tape = [seeded_input("c4x", (B, 6, 16, 16, 64), 52)] + [seeded_input(f"c4n{k}", (B, 6, 16, 16, 64), 53) for k in range(T)]
    avg = torch.tensor([[0.31], [0.12]])

    def make_alignment():
        al = SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid", model_args=dict(V1_ALIGN_ARGS))
        al.model.load_state_dict(seeded_state_dict(al.model.state_dict(), 701))
        al.model.eval()
        return al

    al_cpu = make_alignment()
    buf = {k: torch.as_tensor(v) for k, v in OD.schedule_buffers(OD.beta_schedule("linear", 1000)).items()}
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 32))
    t0 = time.time()
    try:
        def den(z, t, c):
            with torch.no_grad():
                return OU.unet_forward(sd, V1_UNET_CFG, z, t, c)
        traj = OD.ddpm_sample_loop(buf, den, zc, tape, T, align_fn=lambda z, t: al_cpu.get_mean_shift(z, t, avg_x_gt=avg).detach())
    finally:
        torch.set_num_threads(nthr)
    t_cpu = time.time() - t0
    ref = traj[-1]
    errs = {}
    for precision in ("fp32", "fp16x2", "bf16"):
        ldm = _v1_ldm(precision)
        al = make_alignment()
        al.model.cuda()
        ldm.set_alignment(al.get_mean_shift)
        out, inter = ldm.p_sample_loop(cond=zc.cuda(), shape=ldm.get_batch_latent_shape(B), timesteps=T, use_alignment=True,
                                       alignment_kwargs={"avg_x_gt": avg.cuda()}, noise_tape=[x.cuda() for x in tape],
                                       return_intermediates=True, log_every_t=1)
        errs[precision] = rel_l2(out, ref)
        errs[precision + "_by_step"] = [round(rel_l2(inter[k], traj[k]), 6) for k in (1, 25, 50, 75, 100) if k < len(inter)]
        del ldm, al
    print(f"[v1 aligned chain, 100 steps] rel-L2 vs the oracle loop: fp32 {errs['fp32']:.3e} {errs['fp32_by_step']}, "
          f"fp16x2 {errs['fp16x2']:.3e} {errs['fp16x2_by_step']}, bf16 {errs['bf16']:.3e} {errs['bf16_by_step']}; oracle loop {t_cpu:.0f} s on CPU")
    _report("v1_aligned_chain_100", oracle_cpu_s=round(t_cpu, 1), **errs)
    assert errs["fp32"] < 1e-3
    assert errs["fp16x2"] < 1e-3          # the folded-weight engine holds the bar on the guided ancestral chain too (denoiser fp16x2, guidance fp32-class)
    assert errs["bf16"] < 5e-2 and np.isfinite(errs["bf16"])


def test_v1_lane_split_tolerance():
    """V1-size channels, bf16: the split-K Conv3d picks its K slicing from the per-launch batch (<= 16 trajectories), so 8 trajectories
    as one launch, as two lanes of 4 and with split_k = False agree to bf16 noise, not bit for bit; split_k = False is the
    batch-split-reproducible mode (bit-identical across lane counts)."""
    B = 8
    zc, xT = seeded_input("lsc", (B, 7, 16, 16, 64), 61).cuda(), seeded_input("lsx", (B, 6, 16, 16, 64), 62).cuda()
    res = {}
    for split_k in (True, False):
        for lanes in (1, 2):
            ldm = _v1_ldm("bf16")
            ldm.torch_nn_module.split_k = split_k
            ldm.num_streams = lanes
            res[split_k, lanes] = ldm.sample(cond=zc, batch_size=B, sampler="ddim", ddim_steps=2, eta=0.0, x_T=xT, return_decoded=False)
            del ldm
    e_split = rel_l2(res[True, 2], res[True, 1])
    e_mode = rel_l2(res[True, 1], res[False, 1])
    print(f"[v1 lane split, bf16, 8 trajectories, 2 DDIM steps] split-K: 2 lanes vs 1 lane {e_split:.3e}; split-K vs no split {e_mode:.3e}; "
          f"no split: 2 lanes == 1 lane {torch.equal(res[False, 2], res[False, 1])}")
    _report("v1_lane_split", split_vs_lanes=e_split, split_vs_nosplit=e_mode)
    assert torch.equal(res[False, 2], res[False, 1])
    assert e_split < 1e-2 and e_mode < 1e-2
