"""Wall time of one knowledge-alignment guided DDPM step at the v1 configuration (run on the GPU box):
denoiser forward (HIP kernels) | alignment gradient (PyTorch autograd on the alignment network) | step epilogue."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from prediff_amd.presets import V1_ALIGN_ARGS
from prediff_amd.alignment import SEVIRAvgIntensityAlignment

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PRECISION = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # denoiser engine: bf16 | fp8 (e4m3 Conv3d operands) | fp32
GUIDANCE = sys.argv[3] if len(sys.argv) > 3 else "fp32"       # operand form of the guidance network's convolutions: fp32 (hi/lo split) | bf16
dev = torch.device("cuda")
ldm = bench.v1_model(PRECISION, dev)
align = SEVIRAvgIntensityAlignment(guide_scale=50.0, model_args=V1_ALIGN_ARGS, hip_precision=GUIDANCE)
align.model.to(dev)
ldm.set_alignment(align.get_mean_shift)
zc = torch.randn((B, 7, 16, 16, 64), device=dev)
zt = torch.randn(ldm.get_batch_latent_shape(B), device=dev)
t = torch.full((B,), 500, dtype=torch.long, device=dev)
kw = {"avg_x_gt": torch.rand(B, 1, device=dev)}


def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f"B={B}, denoiser precision {PRECISION}, guidance convolutions {GUIDANCE}: denoiser forward (eager) {timed(lambda: ldm.apply_model(zt, t, zc)):.1f} ms; "
      f"alignment gradient {timed(lambda: ldm.alignment_fn(zt, t, zc=zc, y=None, **kw)):.1f} ms; "
      f"p_sample(use_alignment=True) {timed(lambda: ldm.p_sample(zt=zt, zc=zc, t=t, use_alignment=True, alignment_kwargs=kw)):.1f} ms; "
      f"p_sample(no alignment, eager) {timed(lambda: ldm.p_sample(zt=zt, zc=zc, t=t)):.1f} ms")


def loop(graph, n=6):
    ldm.use_hip_graph = graph
    tape = [torch.randn(ldm.get_batch_latent_shape(B)) for _ in range(n + 1)]
    f = lambda: ldm.p_sample_loop(cond=zc, shape=ldm.get_batch_latent_shape(B), use_alignment=True, alignment_kwargs=kw, timesteps=n,
                                  noise_tape=tape)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"aligned DDPM loop, ms/step: eager {loop(False):.1f}")
for ns in (1, 2, 4):
    if B % ns == 0:
        ldm.aligned_lanes = ns
        print(f"aligned DDPM loop, ms/step: denoiser graphs on {ns} lane stream(s) overlapped with the guidance {loop(True):.1f}")
ldm.aligned_lanes = 1
ldm.guidance_high_priority = True
print(f"aligned DDPM loop, ms/step: one denoiser lane, guidance on a HIGH-priority stream {loop(True):.1f} (second run {loop(True):.1f})")
ldm.guidance_high_priority = False
print(f"aligned DDPM loop, ms/step: one denoiser lane, guidance on the caller's stream (again) {loop(True):.1f}")
# history of the guidance gradient at 32 trajectories: 24.9 ms all-PyTorch fp32 (MIOpen Conv3d ~70 %; autocast(bf16) was slower, 26.8 ms)
# -> 15.1 ms with the 3x3x3 convolutions on pd_igemm (_HipConv3d) -> 9.0 ms with the cuboid attention on pd_cuboid_attention(_bwd)
# -> 7.6 ms with GroupNorm -> SiLU -> Conv3d as one row-layout node (pd_groupnorm_silu(_bwd)); all at fp32-class accuracy


# experiment: the guidance gradient (autograd forward + backward) captured in a HIP graph
ldm.use_hip_graph = True
s_zt, s_t = zt.clone(), t.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        ref = ldm.alignment_fn(s_zt, s_t, zc=zc, y=None, **kw)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s_out = ldm.alignment_fn(s_zt, s_t, zc=zc, y=None, **kw)
g.replay(); torch.cuda.synchronize()
print("graphed guidance == eager:", torch.equal(s_out, ref), float((s_out - ref).abs().max()),
      f"; replay {timed(lambda: g.replay()):.2f} ms vs eager {timed(lambda: ldm.alignment_fn(zt, t, zc=zc, y=None, **kw)):.2f} ms")
