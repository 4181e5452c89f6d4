"""Micro-benchmark of the un-fused MFMA attention core (pd_cuboid_attention, bf16) at the level-1 shapes (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.cuboid_geometry import attention_tables
dev = torch.device("cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, shape, Cn, heads, cuboids in ((32, (13, 8, 8), 512, 4, ((13, 1, 1), (1, 8, 1), (1, 1, 8))),
                                    (4, (25, 24, 24), 512, 4, ((25, 1, 1), (1, 24, 1), (1, 1, 24))),
                                    (4, (25, 48, 48), 256, 4, ((25, 1, 1), (1, 48, 1)))):
    ntok = shape[0] * shape[1] * shape[2]
    qkv = torch.randn(B * ntok, 3 * Cn, device=dev).to(torch.bfloat16)
    o = torch.empty(B * ntok, Cn, dtype=torch.bfloat16, device=dev)
    for cuboid in cuboids:
        tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
        vol, nc = tabs["vol"], tabs["nc"]
        bias = torch.randn(heads, vol, vol, device=dev)
        tok = tabs["tok_index"].to(dev)
        t = timed(lambda: L.cuboid_attention(qkv_bf16=qkv, out_bf16=o, tok_index=tok, bias=bias, mask=None, B=B, ntok=ntok, Cn=Cn, heads=heads,
                                             nc=nc, vol=vol, ld_qkv=3 * Cn, ld_out=Cn, scale=(Cn // heads) ** -0.5))
        gb = 4 * B * ntok * Cn * 2 / 1e9
        print(f"B={B} grid {shape} C={Cn} cuboid {cuboid} vol {vol}: {t:8.1f} us  ({gb / t * 1e3:6.2f} TB/s of q,k,v,o bytes)", flush=True)
