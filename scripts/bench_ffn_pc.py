"""ffn64_kernel vs ffn_pc_kernel (producer / consumer waves) at the v1 level-0 FFN shape, interleaved rounds in one process.
argv[1] = pd_ffn_pc_debug_flags (ablations: 1 no W1 DMA after two chunks, 2 no GEMM-1, 4 no activation, 8 no GEMM-2)."""
import ctypes, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.packing import pack_linear, pack_ffn_w2_frag
dbg = ctypes.c_int.in_dll(L.lib(), "pd_ffn_pc_debug_flags")
dbg.value = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.fused_opts(0)
C, Hd = 256, 1024
ROUNDS, REPS = 7, 10
for B in (1, 4, 8, 16, 32, 39):
    M = B * 3328
    x = torch.randn(M, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    w1 = torch.randn(Hd, C, device="cuda") / 16
    w2 = torch.randn(C, Hd, device="cuda") / 128
    w1p, _ = pack_linear(w1, False)
    w2p, _ = pack_linear(w2, False)
    w2f = pack_ffn_w2_frag(w2)
    b1, b2 = torch.zeros(Hd, device="cuda"), torch.zeros(C, device="cuda")
    fns = {"ffn64": lambda: L.ffn_fused(x, x, g, b, w1p, b1, w2p, b2, M, C, Hd), "pc": lambda: L.ffn_fused_pc(x, x, g, b, w1p, b1, w2f, b2, M, C, Hd)}
    ts = {k: [] for k in fns}
    for r in range(ROUNDS):
        for k, fn in fns.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) * 1e3 / REPS)
    print(f"[dbg {dbg.value}] FFN L0 B={B}: " + "  ".join(
        f"{k}: {statistics.median(t):.1f} us (min {min(t):.1f}, {4.0 * M * C * Hd / statistics.median(t) / 1e6:.0f} TF)" for k, t in ts.items()), flush=True)
