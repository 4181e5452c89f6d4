"""Where the guidance gradient's time goes (torch profiler, run on the GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prediff_amd.presets import V1_ALIGN_ARGS
from prediff_amd.alignment import SEVIRAvgIntensityAlignment
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
al = SEVIRAvgIntensityAlignment(guide_scale=50.0, model_args=V1_ALIGN_ARGS)
al.model.to(dev)
zt = torch.randn(B, 6, 16, 16, 64, device=dev)
t = torch.full((B,), 500, dtype=torch.long, device=dev)
kw = {"avg_x_gt": torch.rand(B, 1, device=dev)}
for _ in range(3):
    al.get_mean_shift(zt, t, **kw)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        al.get_mean_shift(zt, t, **kw)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
