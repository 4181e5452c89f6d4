"""bf16 vs e4m3 (scaled MFMA, K = 128) pd_igemm on the long-K launches of the denoiser (run on the GPU box)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prediff_amd import _lib as L
from prediff_amd.packing import pack_conv, pack_conv_fp8, pack_linear, pack_linear_fp8, split_bf16, to_fp8

dev = torch.device("cuda")
DBG = int(os.environ.get("PD_IGEMM_DEBUG", "0"))      # OR-ed into debug_flags (64: the four-phase K-tile of rounds 2-4)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def conv(B, T, H, W, Cin, Cout):
    x = torch.randn(B * T * H * W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / math.sqrt(27 * Cin)
    M = x.shape[0]
    geom = L.conv_geom(B, (T, H, W), (3, 3, 3))
    a16, _ = split_bf16(x, False); w16, _ = pack_conv(w, False)
    a8 = to_fp8(x, 16.0); w8, sw = pack_conv_fp8(w)
    out = torch.empty(M, Cout, device=dev)
    t16 = timed(lambda: L.igemm(a16, w16, M=M, N=Cout, Cin=Cin, taps=27, w_tap_stride=Cout * Cin, geom=geom, out_f32=out, tile=7, debug_flags=DBG))
    t8 = timed(lambda: L.igemm(a8, w8, M=M, N=Cout, Cin=Cin, taps=27, w_tap_stride=Cout * Cin, geom=geom, out_f32=out, alpha=1 / (16 * sw), fp8=True, debug_flags=DBG))
    gf = 2.0 * M * Cout * Cin * 27 / 1e9
    print(f"conv3d B={B} ({T},{H},{W}) {Cin}->{Cout}: M={M}  bf16 {t16:8.1f} us ({gf / t16:6.3f} PF/s)   fp8 {t8:8.1f} us ({gf / t8:6.3f} PF/s)   x{t16 / t8:.2f}", flush=True)


def linear(M, N, K):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / math.sqrt(K)
    a16, _ = split_bf16(x, False); w16, _ = pack_linear(w, False)
    a8 = to_fp8(x, 16.0); w8, sw = pack_linear_fp8(w)
    out = torch.empty(M, N, device=dev)
    t16 = timed(lambda: L.igemm(a16, w16, M=M, N=N, Cin=K, out_f32=out, tile=7, debug_flags=DBG))
    tauto = timed(lambda: L.igemm(a16, w16, M=M, N=N, Cin=K, out_f32=out))
    t8 = timed(lambda: L.igemm(a8, w8, M=M, N=N, Cin=K, out_f32=out, alpha=1 / (16 * sw), fp8=True, debug_flags=DBG))
    gf = 2.0 * M * N * K / 1e9
    print(f"linear {M}x{N}x{K}: bf16 256-tile {t16:8.1f} us ({gf / t16:6.3f} PF/s), auto tile {tauto:8.1f} us   fp8 {t8:8.1f} us ({gf / t8:6.3f} PF/s)   "
          f"x{min(t16, tauto) / t8:.2f} vs the better bf16", flush=True)


conv(32, 13, 16, 16, 256, 256)     # v1 level 0 at 32 trajectories
conv(32, 13, 8, 8, 512, 512)       # v1 level 1
conv(4, 25, 48, 48, 256, 256)      # full resolution level 0, 4 trajectories
conv(4, 25, 24, 24, 512, 512)      # full resolution level 1
linear(8192, 8192, 8192)
linear(106496, 512, 2048)
linear(106496, 768, 256)

print("level-1 linears, v1 at 32 trajectories (M = 26624) and full resolution at 4 (M = 57600); level-0 full resolution (M = 230400)")
for M in (26624, 57600):
    for N, K in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        linear(M, N, K)
linear(230400, 768, 256)
linear(230400, 256, 256)
