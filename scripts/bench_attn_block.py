"""Micro-benchmark of pd_attn_block_fused at the v1 level-0 shapes (run on the GPU box).  argv[1] = ablation flags."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.packing import pack_linear
from prediff_amd.cuboid_geometry import attention_tables
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
OPTS = L.CallOpts(os.environ.get("PD_OPERAND", "bf16"), attn_block_debug_flags=dbg)
B, Cn, heads = (int(sys.argv[3]) if len(sys.argv) > 3 else 32), 256, 4
shape = (13, 16, 16)
ntok = 13 * 16 * 16
for cuboid in ((13, 1, 1), (1, 16, 1)):
    tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    x = torch.randn(B, ntok, Cn, device="cuda")
    g, b = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    wq, _ = pack_linear(torch.randn(3 * Cn, Cn, device="cuda") / 16, False, dtype=OPTS.dtype)
    wp, _ = pack_linear(torch.randn(Cn, Cn, device="cuda") / 16, False, dtype=OPTS.dtype)
    bp = torch.zeros(Cn, device="cuda")
    bias = torch.zeros(heads, vol, vol, device="cuda")
    tok = tabs["tok_index"].cuda()
    out = torch.empty_like(x)
    args = (x, out, g, b, wq, None, wp, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, 0.125)
    for _ in range(3):
        L.attn_block_fused(*args, opts=OPTS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.attn_block_fused(*args, opts=OPTS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = B * (2 * ntok * 768 * 256 + 2 * ntok * 256 * 256 + 4 * ntok * vol * 256) / 1e9
    print(f"[dbg {dbg}] attn_block L0 B={B} cuboid {cuboid}: {us:.1f} us  {gf * 1e3 / us:.1f} TFLOP/s")

# per-phase clock stamps (wave 0 of workgroup 600)
tr = torch.zeros(64, dtype=torch.int64, device="cuda")
OPTS.trace = tr.data_ptr()
L.attn_block_fused(*args, opts=OPTS)
torch.cuda.synchronize()
OPTS.trace = None
t = tr.cpu().tolist()
n = max(i for i, v in enumerate(t) if v) + 1
names = ["start", "tables", "LN", "areg+issue"] + sum([[f"h{h} begin", f"h{h} q,k done", f"h{h} v done", f"h{h} core done"] for h in range(4)], []) + ["proj3+chunk0", "epilogue"]
print("phase durations (shader clocks): " + ", ".join(f"{names[i] if i < len(names) else i}:{t[i] - t[i - 1]}" for i in range(1, n)), " total", t[n - 1] - t[0])
