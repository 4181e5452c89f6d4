"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for bf16 (and the library's own 256 x 256 kernel beside it):
a calibration of the 2.5 PFLOP/s nominal peak the roofline fractions are quoted against -- NOT a product path (measurement only)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L

dev = "cuda"


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in ((4096, 4096, 4096), (8192, 8192, 8192), (106496, 256, 6912), (26624, 512, 13824), (16384, 16384, 4096)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16) / math.sqrt(K)
    out = torch.empty(M, N, device=dev)
    outb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_v = timed(lambda: torch.matmul(a, w.t(), out=outb))
    t_p = timed(lambda: L.igemm(a, w, M=M, N=N, Cin=K, out_f32=out, tile=7))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: torch.matmul (bf16 out) {t_v:8.1f} us = {fl / t_v / 1e6:7.1f} TFLOP/s ({fl / t_v / 1e6 / 2500:.3f} of nominal) | "
          f"pd_igemm 256x256 (fp32 out) {t_p:8.1f} us = {fl / t_p / 1e6:7.1f} TFLOP/s ({fl / t_p / 1e6 / 2500:.3f})", flush=True)
