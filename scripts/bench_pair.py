"""pd_attn_ffn_pair (csrc/pair_block.hip) at the v1 level-0 shapes against the two round-3 kernels it replaces (pd_attn_block_fused +
pd_ffn_fused): agreement and time.  Run on the GPU box:  python scripts/bench_pair.py [B]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.cuboid_geometry import attention_tables
from prediff_amd.packing import pack_linear, pack_pair_block, pack_pair_vecs

DEV = "cuda"
LLL = ("l", "l", "l")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# A/B switches from the environment: PD_OPERAND = bf16 | fp16 (operand type: the pd_f16_* builds), PD_PAIR_NC = form of the units-256 kernel
OPTS = L.CallOpts(os.environ.get("PD_OPERAND", "bf16"), pair_form=int(os.environ.get("PD_PAIR_NC", "0")))
ODT = OPTS.dtype


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    shape, Cn, heads, Hd = (13, 16, 16), 256, 4, 1024
    ntok = shape[0] * shape[1] * shape[2]
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.randn(B, ntok, Cn, generator=g) * 1.5 + 0.2).to(DEV)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    g1, b1n, g2, b2n = 1 + r(Cn, sc=0.1), r(Cn, sc=0.1), 1 + r(Cn, sc=0.1), r(Cn, sc=0.1)
    wqkv, wp = r(3 * Cn, Cn, sc=1 / math.sqrt(Cn)), r(Cn, Cn, sc=1 / math.sqrt(Cn))
    w1, w2 = r(Hd, Cn, sc=1 / math.sqrt(Cn)), r(Cn, Hd, sc=1 / math.sqrt(Hd))
    bp, fb1, fb2 = r(Cn, sc=0.1), r(Hd, sc=0.1), r(Cn, sc=0.1)
    wq_p, wp_p, w1_p, w2_p = (pack_linear(w, False, dtype=ODT)[0] for w in (wqkv, wp, w1, w2))
    ws = pack_pair_block(wqkv, wp, w1, w2, dtype=ODT)
    scale = (Cn // heads) ** -0.5
    for cuboid in ((13, 1, 1), (1, 16, 1), (1, 1, 16)):
        tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
        vol, nc = tabs["vol"], tabs["nc"]
        bias = r(heads, vol, vol, sc=0.5)
        tok = tabs["tok_index"].to(DEV)
        vecs = pack_pair_vecs(g1, b1n, bp, g2, b2n, fb2, fb1, bias)
        assert L.attn_ffn_pair_supported(Cn, heads, Hd, vol)

        def old_attn(t):
            L.attn_block_fused(t, t, g1, b1n, wq_p, None, wp_p, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, scale, tok_affine=tabs["affine"])

        def old_ffn(t):
            L.ffn_fused(t, t, g2, b2n, w1_p, fb1, w2_p, fb2, B * ntok, Cn, Hd, act="gelu")

        def new(t, aff=True):
            L.attn_ffn_pair(t, t, ws, vecs, tok, B, ntok, nc, vol, scale, tok_affine=tabs["affine"] if aff else None, opts=OPTS)

        ref_a = x.clone(); old_attn(ref_a)
        ref_f = x.clone(); old_ffn(ref_f)
        ref_p = ref_a.clone(); old_ffn(ref_p)
        t = x.clone(); new(t); torch.cuda.synchronize()
        print(f"cuboid {cuboid}: rel-L2 of the update vs the round-3 kernels {rel(t - x, ref_p - x):.3e}   finite {bool(torch.isfinite(t).all())}")
        t2 = x.clone(); new(t2, aff=False); torch.cuda.synchronize()
        print(f"   table-driven token ids == affine: {torch.equal(t2, t)};  repeat bit-equal: ", end="")
        t3 = x.clone(); new(t3); torch.cuda.synchronize(); print(torch.equal(t3, t))
        buf = x.clone()
        ta, tf = timeit(lambda: old_attn(buf)), timeit(lambda: old_ffn(buf))
        buf = x.clone()
        tn = timeit(lambda: new(buf))
        gf_a, gf_f = B * 1.7965e9, B * 3.4897e9 * 1.0
        print(f"   B={B}: round-3 attention {ta:.1f} us + FFN {tf:.1f} us = {ta + tf:.1f} us;  pair kernel {tn:.1f} us  "
              f"({(gf_a + gf_f) / tn / 1e6:.0f} TFLOP/s, {(gf_a + gf_f) / tn / 1e6 / 2500:.3f} of peak)")




def level1(B=32):
    """units 512 (the level-1 blocks): pair kernel against an fp32 torch statement of the pair (agreement) and its time."""
    shape, Cn, heads, Hd = (13, 8, 8), 512, 4, 2048
    ntok = shape[0] * shape[1] * shape[2]
    g = torch.Generator(device="cpu").manual_seed(4321)
    x = (torch.randn(B, ntok, Cn, generator=g) * 1.5 + 0.2).to(DEV)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    g1, b1n, g2, b2n = 1 + r(Cn, sc=0.1), r(Cn, sc=0.1), 1 + r(Cn, sc=0.1), r(Cn, sc=0.1)
    wqkv, wp = r(3 * Cn, Cn, sc=Cn ** -0.5), r(Cn, Cn, sc=Cn ** -0.5)
    w1, w2 = r(Hd, Cn, sc=Cn ** -0.5), r(Cn, Hd, sc=Hd ** -0.5)
    bp, fb1, fb2 = r(Cn, sc=0.1), r(Hd, sc=0.1), r(Cn, sc=0.1)
    ws = pack_pair_block(wqkv, wp, w1, w2, dtype=ODT)
    scale = (Cn // heads) ** -0.5
    F = torch.nn.functional
    for cuboid in ((13, 1, 1), (1, 8, 1), (1, 1, 8)):
        tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
        vol, nc = tabs["vol"], tabs["nc"]
        bias = r(heads, vol, vol, sc=0.5)
        tok = tabs["tok_index"].to(DEV)
        vecs = pack_pair_vecs(g1, b1n, bp, g2, b2n, fb2, fb1, bias)
        assert L.attn_ffn_pair_supported(Cn, heads, Hd, vol)
        # fp32 statement
        idx = tok.long().reshape(-1)
        xn = F.layer_norm(x, (Cn,), g1, b1n)[:, idx].reshape(B, nc, vol, Cn)
        qkv = (xn @ wqkv.t()).reshape(B, nc, vol, 3, heads, Cn // heads).permute(3, 0, 1, 4, 2, 5)
        att = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * scale + bias, -1) @ qkv[2]
        o = att.permute(0, 1, 3, 2, 4).reshape(B, nc * vol, Cn) @ wp.t() + bp
        y = x.clone()
        y[:, idx] += o
        ref = y + F.gelu(F.layer_norm(y, (Cn,), g2, b2n) @ w1.t() + fb1) @ w2.t() + fb2
        t = x.clone()
        new = lambda buf, aff=True: L.attn_ffn_pair(buf, buf, ws, vecs, tok, B, ntok, nc, vol, scale, tok_affine=tabs["affine"] if aff else None, units=Cn, opts=OPTS)
        new(t); torch.cuda.synchronize()
        print(f"cuboid {cuboid}: rel-L2 of the update vs fp32 torch {rel(t - x, ref - x):.3e}   finite {bool(torch.isfinite(t).all())}")
        t2 = x.clone(); new(t2, aff=False); t3 = x.clone(); new(t3); torch.cuda.synchronize()
        print(f"   table-driven token ids == affine: {torch.equal(t2, t)};  repeat bit-equal: {torch.equal(t3, t)}")
        buf = x.clone()
        tn = timeit(lambda: new(buf))
        gf = B * ntok * 2.0 * (3 * Cn * Cn + Cn * Cn + 2 * Cn * Hd + 2 * vol * Cn) / 1e9
        print(f"   B={B}: pair kernel {tn:.1f} us  ({gf / tn * 1e3:.0f} TFLOP/s, {gf / tn * 1e3 / 2500:.3f} of peak)")


def trace(B=32):
    """PD_PAIR_DEBUG build: phase clock stamps of wave 0 of workgroup 7 (s_memtime, 100 MHz constant clock on gfx950? printed raw)."""
    import ctypes
    shape, Cn, heads, Hd = (13, 16, 16), 256, 4, 1024
    ntok = 13 * 256
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, ntok, Cn, generator=g).to(DEV)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    ws = pack_pair_block(r(768, 256, sc=1 / 16), r(256, 256, sc=1 / 16), r(1024, 256, sc=1 / 16), r(256, 1024, sc=1 / 32), dtype=ODT)
    tabs = attention_tables(shape, (1, 16, 1), (0, 0, 0), LLL, "zeros")
    vecs = pack_pair_vecs(1 + r(256, sc=.1), r(256, sc=.1), r(256, sc=.1), 1 + r(256, sc=.1), r(256, sc=.1), r(256, sc=.1), r(1024, sc=.1), r(4, 16, 16, sc=.5))
    tok = tabs["tok_index"].to(DEV)
    tr = torch.zeros(256, dtype=torch.int64, device=DEV)
    for _ in range(3):
        L.attn_ffn_pair(x, x, ws, vecs, tok, B, ntok, tabs["nc"], 16, 0.125, tok_affine=tabs["affine"], opts=OPTS)
    L.attn_ffn_pair(x, x, ws, vecs, tok, B, ntok, tabs["nc"], 16, 0.125, tok_affine=tabs["affine"], opts=L.CallOpts(trace=tr.data_ptr()))
    torch.cuda.synchronize()
    t = tr.cpu().tolist()
    n = max(i for i, v in enumerate(t) if v) + 1
    d = [t[i + 1] - t[i] for i in range(n - 1)]
    print("stamps", n, "deltas (s_memtime ticks):", d)
    print("total", t[n - 1] - t[0])


def phases(B=32):
    """Attention phase and FFN phase INSIDE a pair launch (the trace build: prediff_amd/libprediff_hip_trace.so via PD_LIB_PATH; clock
    stamps of wave 0 of workgroup 7 at tile start / attention done / FFN done, three per tile), for the three axial layers of both
    levels at B trajectories.  Prints ONE JSON line: per layer the launch time (HIP events, untraced repeats), the share of the
    workgroup's tile time spent between `tile start` and `attention done` (LayerNorm-1, q/k/v, softmax, P.V, proj, residual) and between
    `attention done` and `FFN done` (LayerNorm-2, FFN-1, GELU, FFN-2, residual), and what each phase's FLOPs over its share of the launch
    come to against the 2.5 PFLOP/s peak.  The workgroup is one of a grid of identical ones: its shares are the launch's."""
    import json
    out = {"trajectories": B, "layers": []}
    for Cn, shape in ((256, (13, 16, 16)), (512, (13, 8, 8))):
        heads, Hd = 4, 4 * Cn
        ntok = shape[0] * shape[1] * shape[2]
        g = torch.Generator(device="cpu").manual_seed(Cn)
        x = torch.randn(B, ntok, Cn, generator=g).to(DEV)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
        ws = pack_pair_block(r(3 * Cn, Cn, sc=Cn ** -0.5), r(Cn, Cn, sc=Cn ** -0.5), r(Hd, Cn, sc=Cn ** -0.5), r(Cn, Hd, sc=Hd ** -0.5), dtype=ODT)
        for cuboid in ((shape[0], 1, 1), (1, shape[1], 1), (1, 1, shape[2])):
            tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
            vol, nc = tabs["vol"], tabs["nc"]
            vecs = pack_pair_vecs(1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), 1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), r(Hd, sc=.1),
                                  r(heads, vol, vol, sc=.5))
            tok = tabs["tok_index"].to(DEV)
            run = lambda o: L.attn_ffn_pair(x, x, ws, vecs, tok, B, ntok, nc, vol, (Cn // heads) ** -0.5, tok_affine=tabs["affine"], units=Cn, opts=o)
            us = timeit(lambda: run(OPTS))
            x.normal_()
            if os.environ.get("PD_PHASES_TIME_ONLY"):     # the same micro-benchmark on the PRODUCT library: the reference duration of the trace build's
                out["layers"].append({"units": Cn, "cuboid": list(cuboid), "launch_us": round(us, 1)})
                continue
            tr = torch.zeros(256, dtype=torch.int64, device=DEV)
            run(L.CallOpts(os.environ.get("PD_OPERAND", "bf16"), pair_form=OPTS.pair_form, trace=tr.data_ptr()))
            torch.cuda.synchronize()
            t = [v for v in tr.cpu().tolist() if v]
            per = 3                                    # stamps per tile (the -DPD_PAIR_TRACE=1 build): tile start, attention done, FFN done
            tiles = len(t) // per
            if tiles == 0 or len(t) % per:
                raise SystemExit(f"{len(t)} clock stamps (want a multiple of {per}): is PD_LIB_PATH the -DPD_PAIR_TRACE=1 build (libprediff_hip_trace.so)?")
            att = sum(t[i * per + 1] - t[i * per] for i in range(tiles))
            ffn = sum(t[i * per + 2] - t[i * per + 1] for i in range(tiles))
            gaps = (t[tiles * per - 1] - t[0]) - att - ffn       # between FFN done of a tile and the next tile's start (row stores / loads)
            sh_a = att / (att + ffn + gaps)
            sh_f = ffn / (att + ffn + gaps)
            gf_att = B * ntok * 2.0 * (3 * Cn * Cn + Cn * Cn + 2 * vol * Cn) / 1e9
            gf_ffn = B * ntok * 2.0 * (2 * Cn * Hd) / 1e9
            out["layers"].append({"units": Cn, "cuboid": list(cuboid), "launch_us": round(us, 1), "tiles_traced": tiles,
                                  "ticks_per_tile": round((att + ffn + gaps) / tiles, 1),
                                  "attention_share": round(sh_a, 4), "ffn_share": round(sh_f, 4),
                                  "attention_gflop": round(gf_att, 2), "ffn_gflop": round(gf_ffn, 2),
                                  "attention_frac_of_peak": round(gf_att / (sh_a * us) * 1e3 / 2500, 4),
                                  "ffn_frac_of_peak": round(gf_ffn / (sh_f * us) * 1e3 / 2500, 4),
                                  "pair_frac_of_peak": round((gf_att + gf_ffn) / us * 1e3 / 2500, 4)})
    print(json.dumps(out))


def ablate(B=32):
    """Time of the pair launch in whatever library PD_LIB_PATH names (scripts/ablate_pair.sh builds -DPD_PAIR_ABLATE variants);
    PD_BENCH_UNITS=512: the level-1 shapes."""
    Cn = int(os.environ.get("PD_BENCH_UNITS", "256"))
    shape, cuboid = ((13, 16, 16), (1, 16, 1)) if Cn == 256 else ((13, 8, 8), (1, 8, 1))
    Hd = 4 * Cn
    ntok = shape[0] * shape[1] * shape[2]
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, ntok, Cn, generator=g).to(DEV)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    ws = pack_pair_block(r(3 * Cn, Cn, sc=Cn ** -0.5), r(Cn, Cn, sc=Cn ** -0.5), r(Hd, Cn, sc=Cn ** -0.5), r(Cn, Hd, sc=Hd ** -0.5), dtype=ODT)
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol = tabs["vol"]
    vecs = pack_pair_vecs(1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), 1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), r(Hd, sc=.1), r(4, vol, vol, sc=.5))
    tok = tabs["tok_index"].to(DEV)
    out = []
    for rep in range(3):
        t = timeit(lambda: L.attn_ffn_pair(x, x, ws, vecs, tok, B, ntok, tabs["nc"], vol, (Cn // 4) ** -0.5, tok_affine=tabs["affine"], units=Cn, opts=OPTS))
        x.normal_()
        out.append(f"{t:.1f} us")
    print(os.environ.get("PD_LIB_PATH", "default"), f"units {Cn}", " | ".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "ablate":
        ablate(int(sys.argv[1]))
    elif len(sys.argv) > 2 and sys.argv[2] == "level1":
        level1(int(sys.argv[1]))
    elif len(sys.argv) > 2 and sys.argv[2] == "phases":
        phases(int(sys.argv[1]))
    elif len(sys.argv) > 2 and sys.argv[2] == "trace":
        trace(int(sys.argv[1]))
    else:
        main()
