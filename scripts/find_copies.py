"""Which torch ops (not pd_* kernels) does one denoiser step launch?  Runs the graph body of bench.py eagerly at B trajectories under
torch.profiler and prints every aten op that launched a device kernel, with its Python call site.  (Run on the GPU box.)"""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
ldm = bench.v1_model("bf16", dev)
z = torch.randn(ldm.get_batch_latent_shape(B), device=dev)
zc = torch.randn((B, 7, 16, 16, 64), device=dev)
t = torch.full((B,), 500, dtype=torch.long, device=dev)
for _ in range(2):
    ldm.apply_model(z, t, zc)
torch.cuda.synchronize()

sites = collections.Counter()
orig = {}


def wrap(name):
    f = getattr(torch.Tensor, name)
    orig[name] = f

    def g(self, *a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        sites[(name, tuple(self.shape), str(self.dtype), " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st)))] += 1
        return f(self, *a, **k)
    setattr(torch.Tensor, name, g)


for n in ("copy_", "contiguous", "float", "to", "clone", "zero_", "fill_"):
    wrap(n)
ldm.apply_model(z, t, zc)
for n, f in orig.items():
    setattr(torch.Tensor, n, f)
for k, v in sites.most_common(40):
    print(v, k)

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ldm.apply_model(z, t, zc)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
