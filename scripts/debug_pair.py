"""PD_PAIR_DEBUG build only: dump the intermediates of head 0 of pd_attn_ffn_pair and compare with a torch statement."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.cuboid_geometry import attention_tables
from prediff_amd.packing import pack_pair_block, pack_pair_vecs
DEV = "cuda"
bf = lambda t: t.to(torch.bfloat16).float()
B, shape, Cn, heads, Hd = 1, (13, 16, 16), 256, 4, 1024
ntok = 13 * 256
g = torch.Generator(device="cpu").manual_seed(1234)
x = (torch.randn(B, ntok, Cn, generator=g) * 1.5 + 0.2).to(DEV)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
g1, b1n, g2, b2n = 1 + r(Cn, sc=0.1), r(Cn, sc=0.1), 1 + r(Cn, sc=0.1), r(Cn, sc=0.1)
wqkv, wp = r(3 * Cn, Cn, sc=1 / 16), r(Cn, Cn, sc=1 / 16)
w1, w2 = r(Hd, Cn, sc=1 / 16), r(Cn, Hd, sc=1 / 32)
bp, fb1, fb2 = r(Cn, sc=0.1), r(Hd, sc=0.1), r(Cn, sc=0.1)
ws = pack_pair_block(wqkv, wp, w1, w2)
cuboid = (1, 16, 1)
tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
vol, nc = tabs["vol"], tabs["nc"]
bias = r(heads, vol, vol, sc=0.5)
tok = tabs["tok_index"].to(DEV)
vecs = pack_pair_vecs(g1, b1n, bp, g2, b2n, fb2, fb1, bias)
lib = L.lib()
dbg = torch.zeros(B * ntok, 256, device=DEV)
ctypes.c_void_p.in_dll(lib, "pd_pair_dbg_buf").value = dbg.data_ptr()
y = bf(torch.nn.functional.layer_norm(x[0], (256,), g1, b1n, 1e-5))
qkv = y @ bf(wqkv).T
rows = tok[0].long()           # tokens of cuboid 0
q0, k0, v0 = qkv[rows, 0:64], qkv[rows, 256:320], qkv[rows, 512:576]
S = bf(q0) @ bf(k0).T * (64 ** -0.5) + bias[0]
P = torch.softmax(S, 1)
O = bf(P) @ bf(v0)
for stage in range(1, 7):
    dbg.zero_()
    ctypes.c_int.in_dll(lib, "pd_pair_dbg_stage").value = stage
    t = x.clone()
    L.attn_ffn_pair(t, t, ws, vecs, tok, B, ntok, nc, vol, 64 ** -0.5, tok_affine=tabs["affine"])
    torch.cuda.synchronize()
    d = dbg[rows]              # [16 tokens][256]
    if stage == 1: print("q", float((d[:, :64] - q0).abs().max()), float(q0.abs().max()))
    if stage == 2: print("k", float((d[:, :64] - k0).abs().max()))
    if stage == 3:
        # cols 0..15: raw s4 (S^T[key 4g+r][query]) -> out[q][4g+r] = S[q][key]; 16..31: logits; 32..47 rb
        print("S raw*scale+bias vs ref", float((d[:, 16:32] - S).abs().max()), " rb vs bias", float((d[:, 32:48] - bias[0]).abs().max()),
              " raw", float((d[:, :16] * 64 ** -0.5 + bias[0] - S).abs().max()))
    if stage == 4: print("P", float((d[:, :16] - P).abs().max()), "finite", bool(torch.isfinite(d).all()), d[0, 16:20].tolist())
    if stage == 5:
        # v plain layout: out[row q][16dt + 4g + r] = V[token 4g+r][feature 16dt + q]
        vv = torch.zeros(16, 64, device=DEV)
        for dt in range(4):
            blk = d[:, 16 * dt:16 * dt + 16]       # [q][token]
            vv[:, 16 * dt:16 * dt + 16] = blk.T
        print("v", float((vv - v0).abs().max()))
    if stage == 6: print("o", float((d[:, :64] - O).abs().max()), "finite", bool(torch.isfinite(d).all()))
    print("   out finite:", bool(torch.isfinite(t).all()))
