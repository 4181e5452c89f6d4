import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.packing import pack_linear
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
use64 = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # 1: 64-row kernel, two workgroups per CU (default); 0: 128-row kernel
OPTS = L.CallOpts(ffn_debug_flags=dbg, ffn_rows128=0 if use64 else 1)
for B in (4, 16, 32):
    M, C, Hd = B * 3328, 256, 1024
    x = torch.randn(M, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    w1, _ = pack_linear(torch.randn(Hd, C, device="cuda") / 16, False)
    w2, _ = pack_linear(torch.randn(C, Hd, device="cuda") / 32, False)
    b1, b2 = torch.zeros(Hd, device="cuda"), torch.zeros(C, device="cuda")
    out = torch.empty_like(x)
    for _ in range(3): L.ffn_fused(x, out, g, b, w1, b1, w2, b2, M, C, Hd, opts=OPTS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.ffn_fused(x, out, g, b, w1, b1, w2, b2, M, C, Hd, opts=OPTS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"[dbg {dbg} use64 {use64}] ffn_fused L0 B={B}: {us:.1f} us  {4.0 * M * C * Hd / us / 1e6:.1f} TFLOP/s")

if use64:
    sys.exit(0)
# per-slot clock stamps of the 128-row kernel (waves 0 and 4 of workgroup 300): work time and barrier wait of every slot
tr = torch.zeros(512, dtype=torch.int64, device="cuda")
OPTS.trace = tr.data_ptr()
L.ffn_fused(x, out, g, b, w1, b1, w2, b2, M, C, Hd, opts=OPTS)
torch.cuda.synchronize()
OPTS.trace = None
t = tr.cpu().tolist()
for grp in (0, 1):
    tt = t[grp * 128:(grp + 1) * 128]
    work = [tt[2 * s] - tt[2 * s - 1] for s in range(1, 34)]
    wait = [tt[2 * s + 1] - tt[2 * s] for s in range(0, 34)]
    pre = t[256 + grp * 64: 256 + grp * 64 + 34]
    dmawait = [tt[2 * s] - pre[s] for s in range(0, 34)]
    print(f"group {grp}: DMA/LDS wait at slot end {dmawait[:12]}")
    print(f"group {grp}: work per slot (clocks) {work[:12]} ...  barrier wait {wait[:12]} ...  loop total {tt[67] - tt[0]}")
