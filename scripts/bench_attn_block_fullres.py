import sys, math, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from prediff_amd import _lib as L
from prediff_amd.cuboid_geometry import attention_tables
from prediff_amd.packing import pack_linear
dev = torch.device("cuda")
B, shape, Cn, heads = 4, (25, 48, 48), 256, 4
ntok = shape[0] * shape[1] * shape[2]
x = torch.randn(B, ntok, Cn, device=dev)
gamma, beta = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev)
wq, _ = pack_linear(torch.randn(3 * Cn, Cn, device=dev) / 16, False)
wp, _ = pack_linear(torch.randn(Cn, Cn, device=dev) / 16, False)
bp = torch.zeros(Cn, device=dev)
def timed(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cuboid in ((25, 1, 1), (1, 48, 1), (1, 1, 48)):
    tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    bias = torch.randn(heads, vol, vol, device=dev)
    tok = tabs["tok_index"].to(dev)
    out = torch.empty_like(x)
    a = torch.empty(B * ntok, Cn, dtype=torch.bfloat16, device=dev)
    qkv = torch.empty(B * ntok, 3 * Cn, dtype=torch.bfloat16, device=dev)
    o = torch.empty(B * ntok, Cn, dtype=torch.bfloat16, device=dev)
    def unfused():
        L.layernorm(x, gamma, beta, a, None, B * ntok, Cn, Cn)
        L.igemm(a, wq, M=B * ntok, N=3 * Cn, Cin=Cn, out_bf16=qkv)
        L.cuboid_attention(qkv_bf16=qkv, out_bf16=o, tok_index=tok, bias=bias, mask=None, B=B, ntok=ntok, Cn=Cn, heads=heads, nc=nc, vol=vol, ld_qkv=3 * Cn, ld_out=Cn, scale=0.125)
        L.igemm(o, wp, M=B * ntok, N=Cn, Cin=Cn, bias=bp, residual=x, out_f32=out)
    def fused():
        L.attn_block_fused(x, out, gamma, beta, wq, None, wp, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, 0.125)
    print(f"cuboid {cuboid} vol {vol}: un-fused chain {timed(unfused):8.1f} us   fused {timed(fused):8.1f} us", flush=True)
