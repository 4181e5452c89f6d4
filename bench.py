#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the PreDiff sampling hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N ...      (outside a launcher: starts N ranks itself, one process per GPU; fewer than N devices = error)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one DDIM step (eta = 0) of the Earthformer-UNet denoiser for a batch of `--batch` independent latent
trajectories per GPU (advanced as `--streams` equal sub-batches = lanes, each a HIP graph on its own stream: independent
trajectories fill each other's idle CUs) at the SEVIR-LR v1 configuration (BASELINE.json configs[1]: latent 6x16x16x64 <- 6x128x128
frames, context 7 frames, 136.8 M-parameter denoiser, bf16 MFMA operands / fp32 accumulate, no knowledge alignment):
denoiser forward + fused step epilogue, replayed from one HIP graph.  Weights are seeded random (no checkpoints
offline), inputs synthetic; everything is resident in HBM before the timed region.  value = trajectory-steps per
second over the whole job = gpus * batch * steps / wall (max over ranks).  Ranks are independent ensemble shards
(weak scaling, no data-path collective; the only exchange of the real sampler is the final all-gather of decoded
frames, outside the step loop).

Two extra objects on the JSON line:
  roofline     - the dominant kernel (Conv3d 3x3x3 implicit GEMM: igemm256_kernel<2,8> where pd_igemm's heuristic picks the
                 256x256 tile, else igemm_kernel<128,128,64,2,false,2,...>): algorithmic FLOPs per launch / its average launch
                 duration measured here with HIP events, against the dense bf16 MFMA peak.
  attention_block - the level-0 cuboid-attention scope, same measurement (second half of BASELINE.json's metric): the (attention, FFN)
                 pair kernel where it runs (FLOPs of both parts over its launch time), with the round-3 attention-block kernel alone beside it.
  cpu_baseline - the oracle (CPU restatement of the reference forward) timed on this box's host cores on a bounded
                 sample of the same workload (kind "port").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF; 2495 TF measured)
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 MFMA peak (scaled K = 128 form; 4.66 PF measured), same guide
UNET_GFLOP_PER_STEP = 653.4        # one denoiser forward, one trajectory (2*MAC)
CONV3D_GFLOP_PER_STEP = 376.9 + 14.9   # 32 TimeEmbedResBlock convs + first_proj
# --config: BASELINE.json configs[1] (the metric's configuration) and configs[4] (full resolution; its fp8 operand path is not
# built: the line says dtype bf16 and is a bring-up measurement of the same engine on the 48x48 latent grid)
WORKLOADS = {
    "v1": dict(unet="V1_UNET_CFG", ldm="V1_LDM_KW", cond=(7, 16, 16, 64), unet_gflop=653.4, conv3d_gflop=376.9 + 14.9, tokens=3328,
               label="SEVIR-LR 7->6 x128x128 (latent 13x16x16, C 256/512, depth [4,4], axial), DDIM-50 eta=0, "
                     "no knowledge alignment (BASELINE.json configs[1])", precision="bf16"),
    "fullres": dict(unet="FULLRES_UNET_CFG", ldm="FULLRES_LDM_KW", cond=(13, 48, 48, 64), unet_gflop=11360.0, conv3d_gflop=6920.0, tokens=57600,
                    label="SEVIR full-res 13->12 x384x384 (latent 25x48x48, C 256/512, depth [4,4], axial cuboids 25/48/48), DDIM-50 "
                          "eta=0 (BASELINE.json configs[4]; fp8 = e4m3 operands for the Conv3d launches and the K >= 512 token linears -- qkv / proj / FFN of "
                          "the level-1 blocks --, bf16 for the fused level-0 blocks)",
                    precision="fp8"),
}
CONV3D_LAUNCHES_PER_STEP = 34
CONV3D_KERNEL_LABEL = "igemm256_kernel<2,8> | igemm_kernel<128,128,64,2,false,2,2,1> per launch (Conv3d 3x3x3 implicit GEMM)"


AB_OPTS = {}        # prediff_amd._lib.CallOpts members set on every denoiser this run builds (--igemm-debug, --pair-form, ...)


def v1_model(precision, device, workload="v1"):
    from prediff_amd import presets
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff_amd.latent_diffusion import LatentDiffusion
    from prediff_amd.seeding import seeded_state_dict
    w = WORKLOADS[workload]
    net = CuboidTransformerUNet(**getattr(presets, w["unet"]), precision=precision)
    for k, v in AB_OPTS.items():          # A/B switches of the command line: per-module options (the library has no process-wide state)
        setattr(net.opts, k, v)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 1234))
    ldm = LatentDiffusion(torch_nn_module=net, first_stage_model=None, cond_stage_model=None, **getattr(presets, w["ldm"]))
    return ldm.to(device).eval()


def pair_phases(B, precision):
    """Attention / FFN phases inside the pair launches at B trajectories: scripts/bench_pair.py phases in a subprocess on the trace build of
    the library (prediff_amd/libprediff_hip_trace.so: the same sources with the pair kernel's clock stamps compiled in -- the product library
    carries none).  None when that build is missing or the run fails (the line then has no `phases`)."""
    import subprocess
    lib = os.path.join(ROOT, "prediff_amd", "libprediff_hip_trace.so")
    if not os.path.exists(lib):
        return None
    env = dict(os.environ, PD_LIB_PATH=lib, PD_OPERAND=precision)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_pair.py"), str(B), "phases"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            return None
        ph = json.loads(r.stdout.strip().splitlines()[-1])
        # the SAME micro-benchmark (same shapes, same stand-alone launches) on the product library: what the trace build's duration is
        # compared with (the in-situ launch time of the forward is another measurement: 225 vs 250 us for one binary)
        envp = {k: v for k, v in env.items() if k != "PD_LIB_PATH"}
        envp["PD_PHASES_TIME_ONLY"] = "1"
        r2 = subprocess.run(cmd, env=envp, capture_output=True, text=True, timeout=240)
        if r2.returncode == 0:
            prod = json.loads(r2.stdout.strip().splitlines()[-1])["layers"]
            for a, b in zip(ph["layers"], prod):
                if (a["units"], a["cuboid"]) == (b["units"], b["cuboid"]):
                    a["product_launch_us"] = b["launch_us"]
        return ph
    except Exception:
        return None


def kernel_times(ldm, B, device, reps=3, cond_shape=(7, 16, 16, 64)):
    """Average duration (s) of (a) the Conv3d implicit-GEMM launches, (b) the level-0 (attention, FFN) pair launches (pd_attn_ffn_pair)
    and (c) the round-3 fused attention-block launches (pd_attn_block_fused: what runs when the pair kernel does not -- measured with
    the pair kernel switched off) of one denoiser forward, with HIP events on the launch stream (eager mode: one event pair per launch).
    Returns (conv_s, conv_launches, attn_s, attn_launches, pair_s, pair_launches, pair512_s, pair512_launches): the pair launches of the
    level-0 (units 256) and of the level-1 (units 512) blocks separately."""
    from prediff_amd import _lib as L
    net = ldm.torch_nn_module
    z = torch.randn(ldm.get_batch_latent_shape(B), device=device)
    zc = torch.randn((B,) + tuple(cond_shape), device=device)
    t = torch.full((B,), 500, dtype=torch.long, device=device)
    orig_igemm, orig_attn, orig_pair = L.igemm, L.attn_block_fused, L.attn_ffn_pair
    conv_pairs, attn_pairs, pair_pairs, pair512_pairs = [], [], [], []

    def bracket(fn, store, a, k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a, **k)
        e1.record()
        store.append((e0, e1))

    def timed_igemm(*a, **k):
        if k.get("taps", 1) == 27:
            bracket(orig_igemm, conv_pairs, a, k)
        else:
            orig_igemm(*a, **k)

    def timed_attn(*a, **k):
        bracket(orig_attn, attn_pairs, a, k)

    def timed_pair(*a, **k):
        bracket(orig_pair, pair512_pairs if k.get("units", 256) == 512 else pair_pairs, a, k)
    net(z, t, zc)
    L.igemm, L.attn_block_fused, L.attn_ffn_pair = timed_igemm, timed_attn, timed_pair
    fuse_pair = getattr(net, "fuse_pair", False)
    try:
        for _ in range(reps):
            net(z, t, zc)
        if pair_pairs:                       # the attention block alone, as the round-3 engine ran it
            L.igemm = orig_igemm
            net.fuse_pair = False
            attn_pairs.clear()
            for _ in range(reps):
                net(z, t, zc)
    finally:
        L.igemm, L.attn_block_fused, L.attn_ffn_pair = orig_igemm, orig_attn, orig_pair
        if hasattr(net, "fuse_pair"):
            net.fuse_pair = fuse_pair
    torch.cuda.synchronize(device)
    avg = lambda ps: sum(a.elapsed_time(b) for a, b in ps) * 1e-3 / len(ps) if ps else None
    return (avg(conv_pairs), len(conv_pairs) // reps, avg(attn_pairs), len(attn_pairs) // reps, avg(pair_pairs), len(pair_pairs) // reps,
            avg(pair512_pairs), len(pair512_pairs) // reps)


def kernel_times_two_lanes(ldm_a, ldm_b, B, device, reps=3, cond_shape=(7, 16, 16, 64)):
    """The Conv3d launches AS THEY RUN IN THE TIMED CONFIGURATION: two lanes of B trajectories advancing concurrently on two streams
    (two module instances driven by two host threads, one HIP-event pair per Conv3d launch on its own stream).  Returns the average
    launch duration (s) over both lanes.  Beside an isolated lane's figure this shows what the two lanes cost each other."""
    import threading
    from prediff_amd import _lib as L
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    pairs = [[], []]
    orig_igemm = L.igemm
    tls = threading.local()

    def timed_igemm(*a, **k):
        lane = getattr(tls, "lane", None)
        if lane is not None and k.get("taps", 1) == 27:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_igemm(*a, **k)
            e1.record()
            pairs[lane].append((e0, e1))
        else:
            orig_igemm(*a, **k)

    inputs = []
    for ldm in (ldm_a, ldm_b):
        z = torch.randn(ldm.get_batch_latent_shape(B), device=device)
        zc = torch.randn((B,) + tuple(cond_shape), device=device)
        t = torch.full((B,), 500, dtype=torch.long, device=device)
        ldm.torch_nn_module(z, t, zc)                      # warm-up: packing, workspaces
        inputs.append((z, t, zc))
    torch.cuda.synchronize(device)
    gate = threading.Barrier(2)

    def run(lane):
        tls.lane = lane
        torch.cuda.set_device(device)
        net = (ldm_a, ldm_b)[lane].torch_nn_module
        with torch.cuda.stream(streams[lane]):
            gate.wait()
            for _ in range(reps):
                net(*inputs[lane])
        streams[lane].synchronize()

    L.igemm = timed_igemm
    try:
        th = [threading.Thread(target=run, args=(l,)) for l in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        L.igemm = orig_igemm
    torch.cuda.synchronize(device)
    allp = pairs[0] + pairs[1]
    return sum(a.elapsed_time(b) for a, b in allp) * 1e-3 / max(1, len(allp)), len(allp)


VAE_ENC_GFLOP_PER_FRAME, VAE_DEC_GFLOP_PER_FRAME = 68.0, 155.2      # SURVEY.md §8(a) a13 / a14 (one 128 x 128 frame, 2 * MAC)


def vae_times(device, trajectories=32, reps=3):
    """Frame-wise KL-VAE of the v1 configuration (the two ends of sample()): encode of the 7 context frames and decode of the 6
    predicted frames of `trajectories` trajectories, bf16 engine, seeded weights; wall time over `reps` calls each."""
    from prediff_amd.autoencoder_kl import AutoencoderKL
    from prediff_amd.presets import V1_VAE_CFG
    from prediff_amd.seeding import seeded_state_dict
    vae = AutoencoderKL(**V1_VAE_CFG, precision="bf16")
    vae.load_state_dict(seeded_state_dict(vae.state_dict(), 77))
    vae = vae.to(device).eval()
    n_enc, n_dec = 7 * trajectories, 6 * trajectories
    x = torch.rand(n_enc, 1, 128, 128, device=device)
    z = torch.randn(n_dec, 64, 16, 16, device=device)
    out = {}
    with torch.no_grad():
        for name, fn, n, gf in (("encode", lambda: vae.encode(x).mode(), n_enc, VAE_ENC_GFLOP_PER_FRAME),
                                ("decode", lambda: vae.decode(z), n_dec, VAE_DEC_GFLOP_PER_FRAME)):
            fn()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(device)
            el = (time.perf_counter() - t0) / reps
            tf = n * gf / el / 1e3
            out[name] = {"frames": n, "ms": round(el * 1e3, 3), "frames_per_s": round(n / el, 1), "gflop_per_frame": gf,
                         "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
    del vae
    return out


def cpu_baseline(budget_s=15.0):
    """Oracle forward (fp32, B=1) on the host cores: bounded sample of the same workload."""
    from oracle import unet as OU
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff_amd.presets import V1_UNET_CFG
    from prediff_amd.seeding import seeded_input, seeded_state_dict
    sd = seeded_state_dict(CuboidTransformerUNet(**V1_UNET_CFG).state_dict(), 1234)     # checkpoint schema (keys, shapes, index buffers)
    x, c = seeded_input("v1x", (1, 6, 16, 16, 64), 2), seeded_input("v1c", (1, 7, 16, 16, 64), 3)
    t = torch.tensor([500])
    # many-core hosts oversubscribe on these small convolutions: time the full core count and a 32-thread run, report the better
    ncpu = torch.get_num_threads()
    best = None
    with torch.no_grad():
        for nthr in sorted({ncpu, min(ncpu, 32)}, reverse=True):
            torch.set_num_threads(nthr)
            OU.unet_forward(sd, V1_UNET_CFG, x, t, c)      # warm-up
            times, t0 = [], time.perf_counter()
            while True:
                t1 = time.perf_counter()
                OU.unet_forward(sd, V1_UNET_CFG, x, t, c)
                times.append(time.perf_counter() - t1)
                el = time.perf_counter() - t0
                if el > budget_s / 2 or len(times) >= 12:
                    break
            # host timing varies +-15 % run to run (other tenants of the box): the value is the best single forward, the spread is reported
            if best is None or 1.0 / min(times) > best[0]:
                best = (1.0 / min(times), nthr, len(times), el, len(times) / el, 1.0 / max(times))
        torch.set_num_threads(ncpu)
    v, nthr, n, el, mean_rate, worst = best
    return {"value": round(v, 4), "unit": "steps/s", "cores": nthr, "kind": "port",
            "mean": round(mean_rate, 4), "worst": round(worst, 4),
            "sample": f"best of {n} oracle denoiser forwards (fp32, B=1, v1 config, torch CPU, {nthr} of {ncpu} threads) in {el:.1f} s; "
                      f"mean {mean_rate:.3f}, slowest {worst:.3f} steps/s"}


def self_launch(n_gpus):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script, one process per GPU, under torch.distributed.run
    (RCCL rendezvous on 127.0.0.1) and return its exit status.  Fewer than N visible devices is an error, never a silent 1-GPU run
    (the reference runs one process per GPU under DDP: scripts/prediff/sevirlr/train_sevirlr_prediff.py:648)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n_gpus and "--launch-check" not in sys.argv:
        print(f"bench.py: --gpus {n_gpus} asked for, {ndev} HIP device(s) visible: refusing to report a smaller job as n_gpus={n_gpus}",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only form the host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def ensemble_e2e(args, ldm, device, rank, world, local_rank, dist):
    """BASELINE configs[2] end to end: ONE context of 7 frames 128 x 128, `--ensemble` members (member k on rank k mod world, seeds by
    member id), per rank: VAE-encode the context, DDIM-50 of its members, VAE-decode; then the one all-gather of the decoded frames
    (prediff_amd/ensemble.py).  `--steps` = repetitions of the whole thing.  value = samples (members) per second, the all-gather alone
    is timed beside it."""
    from prediff_amd.autoencoder_kl import AutoencoderKL
    from prediff_amd.ensemble import all_gather_members, sample_ensemble, shard_members
    from prediff_amd.presets import V1_VAE_CFG
    from prediff_amd.seeding import seeded_state_dict
    if args.config != "v1":
        sys.exit("bench.py --ensemble-e2e: the v1 configuration only")
    vae = AutoencoderKL(**V1_VAE_CFG, precision=args.precision if args.precision in ("bf16", "fp32") else "bf16")
    vae.load_state_dict(seeded_state_dict(vae.state_dict(), 77))
    # the same denoiser instance behind a LatentDiffusion with its first stage (context frames are VAE-encoded by the cond stage =
    # the first stage, as in the reference config: cond_stage_model "__is_first_stage__")
    from prediff_amd import presets
    from prediff_amd.latent_diffusion import LatentDiffusion
    ldm = LatentDiffusion(torch_nn_module=ldm.torch_nn_module, first_stage_model=vae, cond_stage_model="__is_first_stage__",
                          **presets.V1_LDM_KW).to(device).eval()
    E = args.ensemble
    if E < world:
        sys.exit(f"bench.py --ensemble-e2e: --ensemble {E} < {world} ranks")
    y = torch.rand(1, 7, 128, 128, 1, generator=torch.Generator().manual_seed(5)).to(device)
    mine = len(shard_members(E, rank, world))
    ldm.num_streams = lanes_for(mine, args)
    run = lambda: sample_ensemble(ldm, {"y": y}, E, base_seed=1234, sampler="ddim", ddim_steps=50, eta=0.0, force_collective=dist is not None)
    for _ in range(max(1, min(args.warmup, 2))):
        out = run()
    reps = max(1, min(args.steps, 5))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    el = time.perf_counter() - t0
    # the collective alone, on the same payload
    local = out[rank::world].contiguous()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    for _ in range(10):
        all_gather_members(local, E, rank, world, force_collective=dist is not None)
    torch.cuda.synchronize(device)
    ag = (time.perf_counter() - t1) / 10
    if dist is not None:
        te = torch.tensor([el, ag], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        el, ag = float(te[0]), float(te[1])
    if rank == 0:
        assert out.shape == (E, 6, 128, 128, 1) and bool(torch.isfinite(out).all())
        print(json.dumps({
            "metric": "ensemble_samples_per_sec", "value": round(E * reps / el, 3), "unit": "samples/s", "n_gpus": world, "steps": reps,
            "warmup": min(args.warmup, 2), "ms_per_step": round(el / reps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else args.precision,
            "data": "synthetic (seeded random weights, one uniform-random context of 7 frames 128 x 128)",
            "config": {"workload": "SEVIR-LR 7->6 x128x128, ONE ensemble sharded over the GPUs, VAE encode + DDIM-50 + VAE decode + all-gather "
                                   "(BASELINE.json configs[2])", "ensemble": E, "members_per_gpu": mine, "lanes": ldm.num_streams,
                       "parallelism": f"ensemble-shard x{world}"},
            "denoising_steps_per_sec": round(E * 50 * reps / el, 1),
            "all_gather": {"ms": round(ag * 1e3, 4), "payload_bytes_per_rank": int(local.numel() * 4), "backend": "rccl" if dist is not None else "none (one rank)",
                           "fraction_of_sample": round(ag / (el / reps), 6)}}), flush=True)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def lanes_for(batch, args):
    """Lanes for a small per-GPU batch: sub-batches of >= 2 trajectories on concurrent streams (measured, profiles/r02_*sweep*)."""
    if args.small_streams:
        return args.small_streams
    return 2 if batch >= 16 else 1      # round 5 (profiles/r05_r_small_batch_lanes.txt): 8 x 1 / 8 x 2 = 951 / 918 steps/s, 12: 1089 / 1021, 16: 1325 / 1345-1358


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="latent trajectories (ensemble members) per GPU")
    ap.add_argument("--streams", type=int, default=2, help="lanes: the batch advances as this many equal sub-batches on concurrent HIP streams")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "fp16x2", "fp16x2_lin", "fp32", "fp8", "fp8_conv"],
                    help="operand type; default: the one BASELINE.json quotes the workload on (v1: bf16, fullres: fp8)")
    ap.add_argument("--config", default="v1", choices=sorted(WORKLOADS), help="v1 = BASELINE configs[1] (the metric); fullres = configs[4] geometry, bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-test without GPUs: every rank joins a gloo group, rank 0 prints the world it sees, nothing is timed")
    ap.add_argument("--ensemble", type=int, default=32, help="members of the ONE ensemble timed as the strong-scaling line (BASELINE config 3)")
    ap.add_argument("--small-streams", type=int, default=0, help="lanes for the small-batch / strong-scaling lines (0 = automatic)")
    ap.add_argument("--no-extra", action="store_true", help="skip the strong-scaling and small-batch lines")
    ap.add_argument("--ensemble-e2e", action="store_true",
                    help="instead of the step benchmark: time sample_ensemble end to end (BASELINE config 3: ONE context, --ensemble members sharded over "
                         "the ranks, VAE encode, DDIM-50, VAE decode, one all-gather of the decoded frames); reports samples/s and the all-gather time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-ffn", action="store_true")
    ap.add_argument("--no-fused-attn", action="store_true")
    ap.add_argument("--no-pair", action="store_true", help="A/B: the level-0 (attention, FFN) pairs as the two round-3 launches instead of pd_attn_ffn_pair")
    ap.add_argument("--igemm-debug", type=int, default=0, help="A/B: OR-ed into every pd_igemm launch's debug_flags")
    ap.add_argument("--gn-two-launches", action="store_true", help="A/B: GroupNorm as the statistics + apply pair of launches instead of the one-pass kernel")
    ap.add_argument("--pair-form", type=int, default=0, choices=[0, 1, 2, 8],
                    help="A/B: form of pd_attn_ffn_pair at units 256 (1 / 2 = groups per wave with four waves, 8 = eight waves of one group; 0 = automatic)")
    ap.add_argument("--min-k-256", type=int, default=-1, help="A/B: shortest K (taps * Cin) the auto tile choice gives to the 256x256 kernel")
    ap.add_argument("--splitk-max-tiles", type=int, default=-1, help="A/B: split-K Conv3d only for launches of at most this many 256x256 tiles (0 = off)")
    ap.add_argument("--no-tile256", action="store_true", help="A/B: keep pd_igemm on the 128x128 kernel for the long-K launches")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus < 1:
        sys.exit(f"bench.py: --gpus {args.gpus} is not a GPU count")
    if args.gpus != world:
        if "RANK" in os.environ or "LOCAL_RANK" in os.environ:
            # launched by torch.distributed.run with a different process count than asked for: refuse instead of mis-reporting n_gpus
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (--nproc-per-node {args.gpus})")
        sys.exit(self_launch(args.gpus))
    if args.launch_check:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        seen = torch.ones(1)
        dist.all_reduce(seen)
        if args.ensemble_e2e:
            # the sharding + gather logic of the end-to-end line on the gloo world: a stand-in sampler returns each member's id
            from prediff_amd.ensemble import sample_ensemble
            E = args.ensemble
            got = sample_ensemble(None, torch.zeros(1, 7, 4, 4, 1), E, sample_fn=lambda cb, ks: torch.tensor(ks, dtype=torch.float32).reshape(-1, 1, 1, 1, 1).expand(-1, 6, 4, 4, 1))
            ok = bool(torch.equal(got[:, 0, 0, 0, 0], torch.arange(E, dtype=torch.float32)))
            if rank == 0:
                print(json.dumps({"launch_check": True, "ensemble_e2e": True, "n_gpus": world, "ranks_seen": int(seen.item()), "members": E,
                                  "gathered_in_member_order": ok}), flush=True)
        elif rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(seen.item())}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible (there is no CPU path)")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants device {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm; communicator on the current device

    import numpy as np
    from prediff_amd import _lib as L
    from prediff_amd.schedule import make_ddim_sampling_parameters, make_ddim_timesteps
    WL = WORKLOADS[args.config]
    if args.precision is None:
        args.precision = WL["precision"]
    if args.config == "fullres" and args.batch == 64:
        args.batch = 8            # BASELINE config 5: ensemble 64 over 8 GPUs
    B = args.batch
    global CONV3D_KERNEL_LABEL
    if args.igemm_debug:
        AB_OPTS["igemm_debug_or"] = args.igemm_debug
    if args.gn_two_launches:
        AB_OPTS["groupnorm_two_launches"] = 1
    if args.pair_form:
        AB_OPTS["pair_form"] = args.pair_form
    if args.min_k_256 >= 0:
        AB_OPTS["igemm_min_k_256"] = max(args.min_k_256, 1)
    if args.splitk_max_tiles >= 0:
        AB_OPTS["igemm_splitk_max_tiles"] = args.splitk_max_tiles if args.splitk_max_tiles > 0 else -1
    if args.no_tile256:
        CONV3D_KERNEL_LABEL = "igemm_kernel<128,128,64,2,false,2,2,1> (Conv3d 3x3x3 implicit GEMM)"
        AB_OPTS["igemm_disable_256"] = 1
    ldm = v1_model(args.precision, device, args.config)
    if args.no_fused_ffn:
        ldm.torch_nn_module.fuse_ffn = False
    if args.no_fused_attn:
        ldm.torch_nn_module.fuse_attn = False
    if args.no_pair:
        ldm.torch_nn_module.fuse_pair = False
    if args.ensemble_e2e:
        return ensemble_e2e(args, ldm, device, rank, world, local_rank, dist)
    shape = ldm.get_batch_latent_shape(B)
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    zc = torch.randn((B,) + WL["cond"], generator=g).to(device)
    z = torch.randn(shape, generator=g).to(device)

    # DDIM-50 schedule of the reference helpers, cycled if steps > 50
    steps = np.minimum(make_ddim_timesteps("uniform", 50, 1000), 999)
    sig, a, a_prev = make_ddim_sampling_parameters(ldm._alphas_cumprod_f64.astype(np.float32).astype(np.float64), steps, 0.0)
    order = list(reversed(range(len(steps))))
    n_total = max(args.warmup + args.steps, 3 + 20)      # (the strong-scaling / small-batch lines run 3 + <= 20 steps)
    t_all = torch.tensor([[int(steps[order[k % 50]])] * B for k in range(n_total)], dtype=torch.int64, device=device)
    coef_all = torch.tensor([[[a[order[k % 50]], a_prev[order[k % 50]], sig[order[k % 50]]]] * B for k in range(n_total)],
                            dtype=torch.float32, device=device)

    def timed_steps(Bx, Sx, n_steps, n_warm):
        """Advance Bx trajectories per GPU (Sx lanes) for n_warm untimed + n_steps timed DDIM steps; returns (seconds over the timed
        steps: barrier + synchronize on both sides, max over ranks; lanes actually used; final latents)."""
        Sx = Sx if (not args.no_graph and Sx > 1 and Bx % Sx == 0) else 1
        Bl_ = Bx // Sx
        zx, zcx = z[:Bx].contiguous(), zc[:Bx].contiguous()
        ldm.num_streams = Sx
        lanes = st = None
        if args.no_graph:
            out = torch.empty_like(zx)
            noise0 = torch.zeros_like(zx)
        elif Sx > 1:
            lanes = ldm._lanes("ddim", Bx, zcx, device, True)
            for l, lst in enumerate(lanes[0]):
                lst["noise"].zero_()
                lst["z"].copy_(zx[l * Bl_:(l + 1) * Bl_])
        else:
            st = ldm._graph_step("ddim", Bx, zcx, device)
            st["noise"].zero_()
            st["z"].copy_(zx)

        def one_step(k):
            nonlocal zx
            if lanes is not None:
                def fill(lst, sl, k=k):
                    lst["t"].copy_(t_all[k][sl])
                    lst["coef"].copy_(coef_all[k][sl])
                ldm._lane_step(lanes[0], lanes[1], Bl_, device, fill)
            elif st is not None:
                st["t"].copy_(t_all[k][:Bx])
                st["coef"].copy_(coef_all[k][:Bx])
                st["graph"].replay()
                st["z"].copy_(st["out"])
            else:
                eps = ldm.apply_model(zx, t_all[k][:Bx], zcx)
                L.ddim_step(zx, eps, noise0, coef_all[k][:Bx].contiguous(), out, Bx, zx[0].numel())
                zx.copy_(out)

        for k in range(n_warm):
            one_step(k)
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for k in range(n_warm, n_warm + n_steps):
            one_step(k)
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        el = time.perf_counter() - t0
        if lanes is not None:
            for stream in lanes[1]:
                torch.cuda.current_stream(device).wait_stream(stream)
        if dist is not None:
            te = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        fin = torch.cat([lst["z"] for lst in lanes[0]]) if lanes is not None else (st["z"] if st is not None else zx)
        assert bool(torch.isfinite(fin).all()), "non-finite latents after the timed steps"
        return el, Sx

    elapsed, S = timed_steps(B, args.streams, args.steps, args.warmup)
    Bl = B // S

    # ---- BASELINE config 3 as written: ONE ensemble of `--ensemble` members sharded over the GPUs (strong scaling: global work fixed,
    #      per-GPU sub-batch = ensemble / n_gpus), and the small per-GPU batches it implies, reported beside the headline ----
    strong = None
    small = {}
    if not args.no_extra and not args.no_graph and args.config == "v1":
        E = args.ensemble
        k_extra = min(args.steps, 20)
        if E % world == 0 and E // world <= B:
            Be = E // world
            el, Se = timed_steps(Be, lanes_for(Be, args), k_extra, 3)
            strong = {"metric": "denoising_steps_per_sec", "ensemble": E, "trajectories_per_gpu": Be, "lanes": Se, "scaling": "strong",
                      "value": round(E * k_extra / el, 2), "unit": "steps/s", "steps": k_extra, "ms_per_step": round(el / k_extra * 1e3, 4)}
        if world == 1:
            for Bs in (1, 4, 16):
                if Bs <= B:
                    el, Ss = timed_steps(Bs, lanes_for(Bs, args), k_extra, 3)
                    small[f"B{Bs}"] = {"value": round(Bs * k_extra / el, 2), "unit": "steps/s", "lanes": Ss, "ms_per_step": round(el / k_extra * 1e3, 4)}

    # ---- the fp32-class engine (hi/lo split operands, 3 MFMAs per product: the form the <= 1e-3 parity claim is pinned with) at the
    #      headline configuration: its throughput beside the bf16 headline ----
    fp32_line = None
    if not args.no_extra and not args.no_graph and args.config == "v1" and args.precision == "bf16" and world == 1:
        ldm_bf16 = ldm
        ldm = v1_model("fp32", device, args.config)
        k32 = min(args.steps, 8)
        el32, S32 = timed_steps(B, args.streams, k32, 2)
        fp32_line = {"value": round(B * k32 / el32, 2), "unit": "steps/s", "dtype": "bf16x3 (hi/lo split operands, fp32-class accuracy)",
                     "steps": k32, "ms_per_step": round(el32 / k32 * 1e3, 4), "trajectories_per_gpu": B, "lanes": S32,
                     "parity": "v1 DDIM-50 vs the oracle loop 1.4e-5 rel-L2 (tests/test_hip_configs.py::test_v1_ddim50_vs_oracle)"}
        del ldm
        ldm = ldm_bf16
        torch.cuda.empty_cache()

    # ---- the same engine on IEEE-half operands (precision="fp16": 11-bit significands at the bf16 MFMA rate -- the TF32 class of the
    #      reference's own GPU setting): full timed region of the headline configuration ----
    fp16_line = None
    if not args.no_extra and not args.no_graph and args.config == "v1" and args.precision == "bf16" and world == 1:
        ldm_bf16 = ldm
        ldm = v1_model("fp16", device, args.config)
        k16 = min(args.steps, 20)
        el16, S16 = timed_steps(B, args.streams, k16, 3)
        fp16_line = {"value": round(B * k16 / el16, 2), "unit": "steps/s", "dtype": "fp16 operands, fp32 accumulate (TF32-class accuracy)",
                     "steps": k16, "ms_per_step": round(el16 / k16 * 1e3, 4), "trajectories_per_gpu": B, "lanes": S16,
                     "parity": "v1 DDIM-50 vs the oracle loop: tests/test_hip_configs.py::test_v1_ddim50_vs_oracle prints it beside the other modes"}
        del ldm
        ldm = ldm_bf16
        torch.cuda.empty_cache()

    # ---- precision="fp16x2": IEEE-half activations x (hi + lo) IEEE-half weights, two MFMA products per k-step -- the engine that holds the
    #      north-star 1e-3 bar over DDIM-50 with a single-pass activation path (round 6) ----
    fp16x2_line = None
    if not args.no_extra and not args.no_graph and args.config == "v1" and args.precision == "bf16" and world == 1:
        # each in its own process (this script with --precision ... --no-extra): what a user's process running that engine gets, with nothing of
        # the other engines' graphs and workspaces around
        import subprocess

        def own_process(prec):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--precision", prec, "--no-extra", "--no-cpu-baseline", "--steps", str(min(args.steps, 20)),
                                    "--warmup", "5", "--batch", str(B), "--streams", str(args.streams)], capture_output=True, text=True, timeout=600)
                d = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
                return d
            except Exception:
                return None
        dx2, dlin = own_process("fp16x2"), own_process("fp16x2_lin")
        if dx2 is not None:
            fp16x2_line = {"value": dx2["value"], "unit": "steps/s", "dtype": "fp16 activations x fp16 hi+lo weights (two products), fp32 accumulate",
                           "steps": dx2["steps"], "ms_per_step": dx2["ms_per_step"], "trajectories_per_gpu": B, "lanes": dx2["config"].get("lanes"),
                           "measured": "own process (bench.py --precision fp16x2 --no-extra)",
                           "kernels": "folded-weight forms of pd_igemm (w_fold) and of the pair kernel (WP = 2)", "parity": "v1 DDIM-50 vs the oracle loop 3.7e-4, < 1e-3 asserted (tests/test_hip_configs.py::test_v1_ddim50_vs_oracle; "
                                     "its error budget: ::test_v1_fp16_error_budget)"}
            if dlin is not None:
                fp16x2_line["fp16x2_lin"] = {"value": dlin["value"], "ms_per_step": dlin["ms_per_step"],
                                             "what": "the same with the 3x3x3 convolutions on one product (their weight rounding is 3.1e-4 of the 8.8e-4 "
                                                     "weight term of a forward); DDIM-50 5.4e-4, < 1e-3 asserted"}

    if rank == 0:
        n_gpus = world
        value = n_gpus * B * args.steps / elapsed
        ldm.num_streams = S
        ker_s, launches, attn_s, attn_launches, pair_s, pair_launches, pair512_s, pair512_launches = kernel_times(ldm, Bl, device, cond_shape=WL["cond"])     # the kernels as launched: one lane's sub-batch
        flops_per_launch = WL["conv3d_gflop"] * 1e9 * Bl / CONV3D_LAUNCHES_PER_STEP
        achieved = flops_per_launch / ker_s / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "conv3d_hbm_traffic.json")
        if os.path.exists(tp) and args.config == "v1" and args.precision in ("bf16", "fp8", "fp8_conv"):
            try:
                traffic = json.load(open(tp)).get(f"B{Bl}" + ("" if args.precision == "bf16" else "_fp8"))
            except Exception:
                traffic = None
        # --precision fp8 / fp8_conv: the Conv3d launches (the dominant kernel) run on e4m3 operands and are priced against the fp8 peak; every other
        # GEMM of the step stays bf16
        conv_peak = PEAK_FP8_TFLOPS if args.precision.startswith("fp8") else PEAK_BF16_TFLOPS
        line = {
            "metric": "denoising_steps_per_sec", "value": round(value, 2), "unit": "steps/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "fp16", "fp16x2": "fp16x2", "fp16x2_lin": "fp16x2 (Conv3d: one product)", "fp32": "bf16x3", "fp8": "fp8", "fp8_conv": "fp8"}[args.precision],
            "data": "synthetic (seeded random weights of the v1 architecture, N(0,1) latents/context)",
            "config": {"workload": WL["label"],
                       **({"operands": "e4m3 x e4m3 (scaled K=128 MFMA) for the 3x3x3 Conv3d launches and for the K >= 512 linears of the blocks that do not run "
                                       "on the bf16 (attention, FFN) pair kernel (level-1 blocks below ~8 trajectories per launch, cuboid volumes > 16; "
                                       "e4m3 A operands written by LayerNorm, the attention core and the FFN-1 epilogue); bf16 in the pair kernel, "
                                       "the fused level-0 attention / FFN kernels, the K = 256 linears and the VAE"} if args.precision == "fp8" else
                          {"operands": "e4m3 x e4m3 (scaled K=128 MFMA) for the 3x3x3 Conv3d launches, bf16 elsewhere"} if args.precision == "fp8_conv" else {}),
                       "trajectories_per_gpu": B, "global_trajectories": B * n_gpus, "sampler": "ddim50", "hip_graph": not args.no_graph,
                       "lanes": S, "trajectories_per_launch": Bl,
                       "parallelism": f"ensemble-shard x{n_gpus}"},
            "step_tflops": round(WL["unet_gflop"] * 1e9 * value / n_gpus / 1e12, 2),
            "step_frac_of_bf16_peak": round(WL["unet_gflop"] * 1e9 * value / n_gpus / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": {"bound": "mfma", "kernel": CONV3D_KERNEL_LABEL,
                         "achieved": round(achieved, 2), "peak": conv_peak, "unit": "TFLOP/s",
                         "frac": round(achieved / conv_peak, 4), "traffic": traffic,
                         "traffic_source": "profiles/conv3d_hbm_traffic.json: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch from separate "
                                           "rocprofv3 --pmc passes of this kernel (scripts/pmc_bench.sh); not re-measured inside this run",
                         "avg_launch_us": round(ker_s * 1e6, 2), "launches_per_step": launches * S,
                         "gflop_per_launch": round(flops_per_launch / 1e9, 3)},
        }
        if S == 2 and args.config == "v1" and not args.no_extra:
            # the same kernel as it runs in the timed configuration: both lanes active
            ldm_b = v1_model(args.precision, device, args.config)
            ker2_s, n2 = kernel_times_two_lanes(ldm, ldm_b, Bl, device, cond_shape=WL["cond"])
            del ldm_b
            ach2 = flops_per_launch / ker2_s / 1e12
            line["roofline"]["in_situ"] = {"what": "average launch duration with both lanes active (two streams, HIP events per launch)",
                                           "avg_launch_us": round(ker2_s * 1e6, 2), "achieved": round(ach2, 2),
                                           "frac": round(ach2 / conv_peak, 4), "launches_timed": n2}
        if (attn_s or pair_s) and args.config == "v1":
            # level-0 block: LN -> QKV (2*S*3C*C) -> core (4*S*vol*C) -> proj (2*S*C*C), S = 3328 tokens, C = 256, vol 13 or 16;
            # its FFN: 2 * 2*S*C*4C.  
            gf_attn = Bl * (2 * 3328 * 768 * 256 + 2 * 3328 * 256 * 256 + 4 * 3328 * 15 * 256) / 1e9
            gf_ffn = Bl * (2 * 2 * 3328 * 256 * 1024) / 1e9
            attn_alone = None
            if attn_s:
                attn_alone = {"kernel": "attn_block_kernel<256> (LN -> QKV -> cuboid attention -> proj -> +x, level 0; the round-3 kernel, "
                                        "measured with the pair kernel switched off)" if pair_s else
                                        "attn_block_kernel<256> (LN -> QKV -> cuboid attention -> proj -> +x, level 0)",
                              "achieved": round(gf_attn / attn_s / 1e3, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(gf_attn / attn_s / 1e3 / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(attn_s * 1e6, 2),
                              "launches_per_step": attn_launches * S, "gflop_per_launch": round(gf_attn, 3)}
            if pair_s:
                gf = gf_attn + gf_ffn
                line["attention_block"] = {
                    "kernel": "pair_kernel (csrc/pair_block.hip): LN -> QKV -> cuboid attention -> proj -> +x -> LN -> FFN-1 -> GELU -> FFN-2 -> +x "
                              "of one level-0 (attention, FFN) pair in ONE launch, rows register resident",
                    "achieved": round(gf / pair_s / 1e3, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(gf / pair_s / 1e3 / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(pair_s * 1e6, 2),
                    "launches_per_step": pair_launches * S, "gflop_per_launch": round(gf, 3),
                    "gflop_attention_part": round(gf_attn, 3), "gflop_ffn_part": round(gf_ffn, 3),
                    "attention_block_alone_round3_kernel": attn_alone}
            else:
                line["attention_block"] = attn_alone
            if pair512_s:
                # level-1 pair: S = 832 tokens, C = 512, hidden 2048, cuboid volume 13 or 8
                gf1 = Bl * 832 * (2 * 4 * 512 * 512 + 2 * 2 * 512 * 2048 + 4 * 10 * 512) / 1e9
                line["attention_block_level1"] = {
                    "kernel": "pair_kernel<1, 2> (csrc/pair_block.hip): the same pair at units 512 (4 heads of 128, hidden 2048), 16 rows per wave; "
                              "replaces qkv / proj / FFN-1 / FFN-2 GEMM launches + 2 LayerNorm + attention core launches",
                    "achieved": round(gf1 / pair512_s / 1e3, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(gf1 / pair512_s / 1e3 / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(pair512_s * 1e6, 2),
                    "launches_per_step": pair512_launches * S, "gflop_per_launch": round(gf1, 3)}
        if pair_s and args.config == "v1" and not args.no_extra and n_gpus == 1 and args.precision in ("bf16", "fp16"):
            ph = pair_phases(Bl, args.precision)
            if ph is not None:
                for key, units in (("attention_block", 256), ("attention_block_level1", 512)):
                    rows = [r for r in ph["layers"] if r["units"] == units]
                    if key in line and rows:
                        mean = lambda k: sum(r[k] for r in rows) / len(rows)
                        sh_a, sh_f = mean("attention_share"), mean("ffn_share")
                        # FLOPs of the two phases per launch over their share of THIS run's launch time
                        if units == 256:
                            gfa, gff = line[key]["gflop_attention_part"], line[key]["gflop_ffn_part"]
                        else:
                            gfa = Bl * 832 * (2 * 4 * 512 * 512 + 4 * 10 * 512) / 1e9
                            gff = Bl * 832 * (2 * 2 * 512 * 2048) / 1e9
                        us = line[key]["avg_launch_us"]
                        tb_us = mean("launch_us")
                        pr_us = mean("product_launch_us") if all("product_launch_us" in r for r in rows) else us
                        if abs(tb_us / pr_us - 1.0) > 0.10:
                            # the stamps are evidence for the product kernel only if the traced build runs like it (the old 24-stamp
                            # build of the units-512 form ran 3x slower): say so instead of printing shares of a different kernel
                            line[key]["phases"] = {"dropped": "trace build launch differs from the product launch by more than 10 %",
                                                   "trace_build_launch_us": round(tb_us, 1), "product_launch_us_same_microbench": round(pr_us, 1)}
                            continue
                        line[key]["phases"] = {
                            "what": "the launch split by the kernel's own clock stamps (three per tile, held in scalar registers: the -DPD_PAIR_TRACE=1 build "
                                    "of the same source, scripts/bench_pair.py phases, mean over the three axial layers; applied to this run's launch time): "
                                    "attention = LayerNorm-1 .. proj + residual, ffn = LayerNorm-2 .. FFN-2 + residual",
                            "attention_share_of_launch": round(sh_a, 4), "ffn_share_of_launch": round(sh_f, 4),
                            "attention_us": round(sh_a * us, 2), "ffn_us": round(sh_f * us, 2),
                            "attention_gflop": round(gfa, 3), "ffn_gflop": round(gff, 3),
                            "attention_frac_of_peak": round(gfa / (sh_a * us) * 1e3 / PEAK_BF16_TFLOPS, 4),
                            "ffn_frac_of_peak": round(gff / (sh_f * us) * 1e3 / PEAK_BF16_TFLOPS, 4),
                            "trace_build_launch_us": round(tb_us, 1), "product_launch_us_same_microbench": round(pr_us, 1),
                            "trace_build_vs_product": round(tb_us / pr_us, 3)}
        if strong is not None:
            if n_gpus == 1 and "B4" in small and strong["ensemble"] == 32:
                # what the SAME ensemble would do on 8 GPUs (4 members each): the step loop has no collective, so 8 x the measured
                # 4-trajectory rate -- a projection from this GPU's own numbers, not a measurement
                strong["projected_8gpu"] = {"trajectories_per_gpu": 4, "steps_per_sec": round(8 * small["B4"]["value"], 1),
                                            "speedup_vs_this_gpu": round(8 * small["B4"]["value"] / strong["value"], 2),
                                            "basis": "8 x small_batch.B4 (no collective in the step loop); north-star target 6x",
                                            # the arithmetic of the 6x target: what 4 trajectories per GPU would have to
                                            # run at, and what they would run at if every kernel kept the per-trajectory cost it has at
                                            # full occupancy (this run's headline rate) -- the ceiling of any small-batch engine
                                            "needed_B4_steps_per_sec_for_6x": round(6.0 * strong["value"] / 8.0, 1),
                                            "B4_at_full_occupancy_kernel_efficiency": round(value, 1),
                                            "speedup_ceiling_at_that_efficiency": round(8.0 * value / strong["value"], 2),
                                            "B4_fraction_of_that_ceiling": round(small["B4"]["value"] / value, 3)}
            line["ensemble_strong_scaling"] = strong      # BASELINE configs[2]: ensemble=32 over the node's GPUs
        if small:
            line["small_batch"] = small                   # B in {1..16} beside the headline batch
        if fp32_line is not None:
            line["precision_fp32"] = fp32_line
        if fp16_line is not None:
            line["precision_fp16"] = fp16_line
        if fp16x2_line is not None:
            line["precision_fp16x2"] = fp16x2_line
        if not args.no_extra and args.config == "v1" and n_gpus == 1:
            line["vae"] = vae_times(device, trajectories=min(B, 32))
        if not args.no_cpu_baseline and n_gpus == 1 and args.config == "v1":
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
