"""Experiment (round 2): does the PyTorch-autograd guidance gradient get faster with channels_last_3d Conv3d weights / MIOpen
find mode?  (run on the GPU box)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prediff_amd.presets import V1_ALIGN_ARGS
from prediff_amd.alignment import SEVIRAvgIntensityAlignment

B = 32
dev = torch.device("cuda")
zt = torch.randn(B, 6, 16, 16, 64, device=dev)
t = torch.full((B,), 500, dtype=torch.long, device=dev)
kw = {"avg_x_gt": torch.rand(B, 1, device=dev)}


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for cl in (False, True):
        al = SEVIRAvgIntensityAlignment(guide_scale=50.0, model_args=V1_ALIGN_ARGS)
        al.model.to(dev)
        if cl:
            for m in al.model.modules():
                if isinstance(m, torch.nn.Conv3d):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last_3d)
        ref = al.get_mean_shift(zt, t, **kw)
        print(f"cudnn.benchmark={bench_mode} channels_last_3d={cl}: guidance gradient {timed(lambda: al.get_mean_shift(zt, t, **kw)):.2f} ms", flush=True)
