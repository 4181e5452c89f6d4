"""Debug helper (GPU box): localise a mismatch of pd_attn_block_fused against the un-fused chain."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.packing import pack_linear
from prediff_amd.cuboid_geometry import attention_tables
DEV = "cuda"
shape, cuboid, Cn, heads, B = (3, 5, 6), (3, 1, 1), 128, 2, 1
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shape, cuboid, Cn, heads, B = (13, 16, 16), (1, 16, 1), 256, 4, 1
T, H, W = shape
ntok = T * H * W
g = torch.Generator().manual_seed(0)
x = torch.randn(B, ntok, Cn, generator=g).to(DEV)
gamma, beta = torch.ones(Cn, device=DEV), torch.zeros(Cn, device=DEV)
wqkv = (torch.randn(3 * Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV)
wp = (torch.randn(Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV)
bp = torch.randn(Cn, generator=g).to(DEV)
tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
vol, nc = tabs["vol"], tabs["nc"]
bias = torch.zeros(heads, vol, vol, device=DEV)
tok = tabs["tok_index"].to(DEV)
scale = (Cn // heads) ** -0.5


def run(wq, wpm, tag, bq=None):
    wq_p, _ = pack_linear(wq, False)
    wp_p, _ = pack_linear(wpm, False)
    a = torch.empty(B * ntok, Cn, dtype=torch.bfloat16, device=DEV)
    L.layernorm(x, gamma, beta, a, None, B * ntok, Cn, Cn)
    qkv = torch.empty(B * ntok, 3 * Cn, dtype=torch.bfloat16, device=DEV)
    L.igemm(a, wq_p, M=B * ntok, N=3 * Cn, Cin=Cn, bias=bq, out_bf16=qkv)
    o = torch.zeros(B * ntok, Cn, dtype=torch.bfloat16, device=DEV)
    L.cuboid_attention(qkv_bf16=qkv, out_bf16=o, tok_index=tok, bias=bias, mask=None, B=B, ntok=ntok, Cn=Cn, heads=heads, nc=nc,
                       vol=vol, ld_qkv=3 * Cn, ld_out=Cn, scale=scale)
    ref = torch.empty_like(x)
    L.igemm(o, wp_p, M=B * ntok, N=Cn, Cin=Cn, bias=bp, residual=x, out_f32=ref)
    out = torch.full_like(x, float("nan"))
    L.attn_block_fused(x, out, gamma, beta, wq_p, bq, wp_p, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, scale)
    torch.cuda.synchronize()
    d = (out - ref)
    nan_rows = (~torch.isfinite(out)).any(-1).sum().item()
    fin = torch.isfinite(d)
    print(f"{tag}: nan rows {nan_rows}/{B * ntok}, max|d| over finite {d[fin].abs().max().item() if fin.any() else -1:.4g}, "
          f"|ref-x| max {(ref - x).abs().max().item():.3g}")
    bad = (d.abs() > 1e-2) | ~torch.isfinite(d)
    if bad.any():
        rows = bad.any(-1).nonzero()[:8, 1].tolist()
        cols = bad.any(1)[0].nonzero()[:16, 0].tolist()
        print("   first bad rows", rows, "bad cols", cols, "n bad cols", int(bad.any(1)[0].sum()))


run(wqkv, torch.zeros_like(wp), "Wp=0 (epilogue/gather only)")
wv_only = wqkv.clone(); wv_only[:2 * Cn] = 0
run(wv_only, wp, "q=k=0 (uniform softmax)")
run(wqkv, wp, "full")
run(torch.zeros_like(wqkv), wp, "Wqkv=0 (O must be 0)")
wq_only = wqkv.clone(); wq_only[Cn:] = 0
run(wq_only, wp, "only q nonzero (k=v=0 -> O=0)")
wk_only = wqkv.clone(); wk_only[:Cn] = 0; wk_only[2 * Cn:] = 0
run(wk_only, wp, "only k nonzero (O=0)")

bq1 = torch.zeros(3 * Cn, device=DEV); bq1[2 * Cn:] = torch.arange(Cn, device=DEV) * 0.01 + 1.0
run(torch.zeros_like(wqkv), wp, "v = bias only (per-d constant)", bq1)
wv_id = torch.zeros_like(wqkv); wv_id[2 * Cn:] = torch.eye(Cn, device=DEV)
run(wv_id, wp, "Wv = identity (v = LN(x))")
