#!/usr/bin/env python3
"""Micro-benchmark of pd_igemm on the shapes of one v1 denoiser forward (run on the GPU box).
   python scripts/bench_igemm.py [--batch 16] [--reps 20] [--only conv3d_l0]"""
import argparse
import os
import sys
import math

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--debug", type=int, default=0)
ap.add_argument("--split", action="store_true", help="hi/lo split operands (the precision=\"fp32\" engine): 3 products per fragment pair")
args = ap.parse_args()
B = args.batch
dev = "cuda"


def run(name, M, N, K, taps=1, geom=None, out_bf16=False, act="none", residual=False):
    Kp = K
    a = (torch.randn(M if taps == 1 else geom["B"] * geom["Ti"] * geom["Hi"] * geom["Wi"], Kp, device=dev)).to(torch.bfloat16)
    w = (torch.randn(taps, N, Kp, device=dev) / math.sqrt(K * taps)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    of = None if out_bf16 else torch.empty(M, N, device=dev)
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if out_bf16 else None
    res = torch.randn(M, N, device=dev) if residual else None
    kw = dict(M=M, N=N, Cin=Kp, taps=taps, w_tap_stride=N * Kp, geom=geom, bias=bias, act=act, residual=res, out_f32=of, out_bf16=ob, tile=args.tile, debug_flags=args.debug)
    if args.split:        # both halves of an operand in one allocation, as the engine keeps them
        a2, w2 = torch.empty((2,) + tuple(a.shape), dtype=torch.bfloat16, device=dev), torch.empty((2,) + tuple(w.shape), dtype=torch.bfloat16, device=dev)
        a2[0].copy_(a); a2[1].copy_(a * 0.004); w2[0].copy_(w); w2[1].copy_(w * 0.004)
        a, w = a2[0], w2[0]
        kw.update(A_lo=a2[1], W_lo=w2[1])
        if ob is not None:
            kw.update(out_bf16_lo=torch.empty_like(ob))
    for _ in range(3):
        L.igemm(a, w, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        L.igemm(a, w, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    fl = 2.0 * M * N * K * taps
    print(f"{name:28s} M={M:6d} N={N:5d} K={K * taps:6d}  {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)


S0, S1 = 13 * 16 * 16, 13 * 8 * 8
cases = {
    "conv3d_l0": lambda: run("conv3d_l0 256->256", B * S0, 256, 256, 27, L.conv_geom(B, (13, 16, 16), (3, 3, 3)), residual=True),
    "conv3d_l1": lambda: run("conv3d_l1 512->512", B * S1, 512, 512, 27, L.conv_geom(B, (13, 8, 8), (3, 3, 3)), residual=True),
    "qkv_l0": lambda: run("qkv_l0 256->768", B * S0, 768, 256, out_bf16=True),
    "proj_l0": lambda: run("proj_l0 256->256", B * S0, 256, 256, residual=True),
    "ffn1_l0": lambda: run("ffn1_l0 256->1024 gelu", B * S0, 1024, 256, out_bf16=True, act="gelu"),
    "ffn2_l0": lambda: run("ffn2_l0 1024->256", B * S0, 256, 1024, residual=True),
    "qkv_l1": lambda: run("qkv_l1 512->1536", B * S1, 1536, 512, out_bf16=True),
    "proj_l1": lambda: run("proj_l1 512->512", B * S1, 512, 512, residual=True),
    "ffn1_l1": lambda: run("ffn1_l1 512->2048 gelu", B * S1, 2048, 512, out_bf16=True, act="gelu"),
    "ffn2_l1": lambda: run("ffn2_l1 2048->512", B * S1, 512, 2048, residual=True),
    "gemm_4k": lambda: run("gemm 4096^3", 4096, 4096, 4096),
}
for k, f in cases.items():
    if not args.only or k in args.only.split(","):
        f()
