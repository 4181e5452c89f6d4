"""ctypes binding of libprediff_hip.so (include/prediff_hip.h) + thin tensor-level wrappers.

PyTorch is used here only as the owner of device memory and streams: every wrapper passes raw
device pointers and the current HIP stream to the C ABI.  There is NO fallback: if the shared
library is missing or a kernel reports an error, an exception is raised.
"""
import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PD_LIB_PATH: another build of the same library (A/B of compiler flags); the default is the in-tree build
LIB_PATH = os.environ.get("PD_LIB_PATH") or os.path.join(_HERE, "libprediff_hip.so")

ABI_VERSION = 4          # pd_abi_version() of the library this binding was written for

ACT = {"none": 0, None: 0, "identity": 0, "gelu": 1, "silu": 2, "leaky": 3, "relu": 4}


class PrediffHipError(RuntimeError):
    pass


class IgemmArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("A", "A_lo", "W", "W_lo", "bias", "rowvec", "residual", "mul", "out_f32", "out_bf16", "out_bf16_lo")] + \
               [(n, C.c_int64) for n in
                ("a_batch_stride", "w_batch_stride", "out_batch_stride", "outb_batch_stride", "res_batch_stride",
                 "w_tap_stride")] + \
               [(n, C.c_int32) for n in
                ("nbatch", "M", "N", "Cin", "taps", "lda", "ldw", "B", "Ti", "Hi", "Wi", "To", "Ho", "Wo",
                 "KT", "KH", "KW", "st", "sh", "sw", "pt", "ph", "pw", "ut", "uh", "uw", "vT", "vH", "vW",
                 "rows_per_sample", "ld_rowvec", "ld_res", "res_period", "ld_mul", "act", "ld_out", "ld_outb", "split")] + \
               [("alpha", C.c_float), ("tile", C.c_int32), ("vec_epilogue", C.c_int32), ("a_bytes", C.c_uint32), ("w_bytes", C.c_uint32), ("debug_flags", C.c_int32), ("ksplit", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_ws_elems", C.c_int64), ("fp8", C.c_int32), ("out_fp8_log2", C.c_int32),
                ("operand", C.c_int32), ("disable_256", C.c_int32), ("min_k_256", C.c_int32), ("splitk_max_tiles", C.c_int32),
                ("w_fold", C.c_int32), ("reserved0", C.c_int32)]


class CuboidAttnArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("qkv_bf16", "qkv_f32", "tok_index", "bias", "mask", "out_bf16", "out_bf16_lo", "out_f32")] + \
               [(n, C.c_int32) for n in ("B", "ntok", "C", "heads", "nc", "vol", "ld_qkv", "ld_out")] + \
               [("scale", C.c_float), ("force_generic", C.c_int32), ("out_fp8_log2", C.c_int32), ("tok_out", C.c_void_p),
                ("operand", C.c_int32), ("qkv_fp8_log2", C.c_int32)]


OPERAND = {"bf16": 0, "fp16": 1}            # enum pd_operand


class CallOpts(C.Structure):
    """pd_call_opts: the per-call options of the library (operand type + A/B switches + profiling hooks).  A module owns one instance and
    passes it on every launch: nothing is process-global.  All-zero = bfloat16 operands, production settings.  `igemm_*` are host-side
    defaults the `igemm` wrapper copies into pd_igemm_args (the library's own struct has no such members)."""
    _fields_ = [(n, C.c_int32) for n in ("operand", "attn_block_table_ids", "ffn_rows128", "groupnorm_two_launches", "pair_form",
                                         "ffn_debug_flags", "attn_block_debug_flags", "small_grid", "w_fold", "reserved1")] + [("trace", C.c_void_p)]

    def __init__(self, operand="bf16", **kw):
        super().__init__()
        self.operand = OPERAND[operand] if isinstance(operand, str) else int(operand)
        # pd_igemm_args members a caller may preset for every pd_igemm launch made with these options (A/B switches of bench.py / scripts)
        self.igemm_tile = self.igemm_debug_or = self.igemm_disable_256 = self.igemm_min_k_256 = self.igemm_splitk_max_tiles = 0
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def dtype(self):
        """torch dtype of the 16-bit operand buffers these options describe"""
        return torch.float16 if self.operand == OPERAND["fp16"] else torch.bfloat16

    _HOST_FIELDS = ("igemm_tile", "igemm_debug_or", "igemm_disable_256", "igemm_min_k_256", "igemm_splitk_max_tiles")

    def _state(self):
        """every member but `trace` (a device pointer of a profiling run: meaningless in a copy or another process)"""
        st = {n: getattr(self, n) for n, _ in self._fields_ if n != "trace"}
        st.update({n: getattr(self, n) for n in self._HOST_FIELDS})
        return st

    def replace(self, **kw):
        """a copy with some members changed (the options a module passes are never mutated per launch)"""
        st = self._state()
        st["trace"] = self.trace
        st.update(kw)
        return CallOpts(**st)

    def __reduce__(self):
        # ctypes refuses to pickle structures holding pointers; a module that owns a CallOpts must stay deepcopy- / pickle-able
        # (EMA by deepcopy, DDP spawn, torch.save(module))
        return (_callopts_from_state, (self._state(),))


def _callopts_from_state(st):
    return CallOpts(**st)


def _opts_ref(opts):
    return C.byref(opts) if opts is not None else None


_lib = None

_PROTOS = {
    "pd_abi_version": (C.c_int, []),
    "pd_last_error": (C.c_char_p, []),
    "pd_sizeof_igemm_args": (C.c_int, []),
    "pd_sizeof_cuboid_attn_args": (C.c_int, []),
    "pd_sizeof_call_opts": (C.c_int, []),
    "pd_igemm": (C.c_int, [C.POINTER(IgemmArgs), C.c_void_p]),
    "pd_layernorm": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_float, C.POINTER(CallOpts), C.c_void_p]),
    "pd_layernorm_fp8": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "pd_patch_merge_layernorm": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 9 + [C.c_float, C.c_void_p]),
    "pd_patch_merge_layernorm_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 9 + [C.c_float, C.c_int, C.POINTER(CallOpts), C.c_void_p]),
    "pd_groupnorm_nchunk": (C.c_int, [C.c_int, C.c_int]),
    "pd_groupnorm_silu": (C.c_int, [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 5 +
                          [C.c_float, C.c_int, C.POINTER(CallOpts), C.c_void_p]),
    "pd_groupnorm_stats": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "pd_conv2d_gn_silu_supported": (C.c_int, [C.c_int] * 5),
    "pd_conv2d_gn_silu": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 6 + [C.POINTER(CallOpts), C.c_void_p]),
    "pd_conv2d_up2": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.POINTER(CallOpts), C.c_void_p]),
    "pd_groupnorm_silu_fp8": (C.c_int, [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "pd_groupnorm_silu_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]),
    "pd_cast_rows": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int] * 6 + [C.POINTER(CallOpts), C.c_void_p]),
    "pd_cuboid_attention": (C.c_int, [C.POINTER(CuboidAttnArgs), C.c_void_p]),
    "pd_cuboid_attention_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 9 + [C.c_float, C.c_void_p]),
    "pd_softmax_rows": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int] * 3 + [C.POINTER(CallOpts), C.c_void_p]),
    "pd_unet_build_input": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]),
    "pd_timestep_embedding": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p]),
    "pd_linear_small": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "pd_add_rowtable": (C.c_int, [C.c_void_p] * 2 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pd_add": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "pd_ddpm_step": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "pd_ddim_step": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_void_p]),
    "pd_nchw_to_nhwc": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "pd_nhwc_to_nchw": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "pd_ffn_fused_supported": (C.c_int, [C.c_int, C.c_int]),
    "pd_attn_block_fused_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pd_attn_block_fused": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_void_p]),
    "pd_attn_block_fused_ex": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_void_p, C.POINTER(CallOpts), C.c_void_p]),
    "pd_ffn_fused": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(CallOpts), C.c_void_p]),
    "pd_attn_ffn_pair_supported": (C.c_int, [C.c_int] * 5),
    "pd_attn_ffn_pair_cuboids_per_group": (C.c_int, [C.c_int]),
    "pd_attn_ffn_pair": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float] * 3 + [C.POINTER(CallOpts), C.c_void_p]),
    "pd_attn_ffn_pair_split_ws_floats": (C.c_int64, [C.c_int] * 3),
    "pd_attn_ffn_pair_split": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_float] * 3 + [C.c_void_p, C.c_int64, C.POINTER(CallOpts), C.c_void_p]),
    "pd_ffn_rows_supported": (C.c_int, [C.c_int] * 3),
    "pd_ffn_rows": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_float, C.POINTER(CallOpts), C.c_void_p]),
    "pd_sevir_skill_counts": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PrediffHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C prediff_amd/csrc`. prediff_amd has no non-HIP execution path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            f = getattr(l, name)      # AttributeError if a declared symbol is missing
            f.restype = res
            f.argtypes = args
        if (l.pd_sizeof_igemm_args() != C.sizeof(IgemmArgs) or l.pd_sizeof_cuboid_attn_args() != C.sizeof(CuboidAttnArgs)
                or l.pd_sizeof_call_opts() != C.sizeof(CallOpts) or l.pd_abi_version() != ABI_VERSION):
            raise PrediffHipError(f"{LIB_PATH} is stale: its ABI version / argument structs do not match prediff_amd/_lib.py "
                                  f"(rebuild with `make -C prediff_amd/csrc`)")
        _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise PrediffHipError(f"{what} failed (status {rc}): {lib().pd_last_error().decode()}")


def stream_ptr() -> int:
    """The current HIP stream of the current device.  Every module entry point (forward / encode / decode / sample / update) runs
    inside `on_device(x)`, which makes the operands' device the current one, so this is the stream of the tensors' device."""
    return torch.cuda.current_stream().cuda_stream


def on_device(t: torch.Tensor):
    """Context manager: make `t`'s device current (kernels are launched on the current stream of the current device; a module on
    cuda:1 called while cuda:0 is current must not launch on a device-0 stream with device-1 pointers)."""
    if not t.is_cuda:
        raise PrediffHipError("prediff_amd kernels need CUDA(HIP) tensors; got a CPU tensor")
    return torch.cuda.device(t.device)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor, dtype=None):
    if not t.is_cuda:
        raise PrediffHipError("prediff_amd kernels need CUDA(HIP) tensors; got a CPU tensor")
    if dtype is not None and t.dtype != dtype:
        raise PrediffHipError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise PrediffHipError("expected a contiguous tensor")
    return t


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# --------------------------------------------------------------------------------------------------
# wrappers
# --------------------------------------------------------------------------------------------------
def igemm(A, W, *, M, N, Cin, lda=None, ldw=None, taps=1, w_tap_stride=0, geom=None,
          A_lo=None, W_lo=None, bias=None, rowvec=None, rows_per_sample=0, residual=None, res_period=0,
          ld_res=None, mul=None, act="none", alpha=1.0, out_f32=None, out_bf16=None, out_bf16_lo=None,
          ld_out=None, ld_outb=None, nbatch=1, a_batch_stride=0, w_batch_stride=0, out_batch_stride=0,
          outb_batch_stride=0, res_batch_stride=0, tile=0, debug_flags=0, splitk_ws=None, fp8=False, out_fp8_log2=0, opts=None):
    """Thin wrapper around pd_igemm.  `geom` = dict(B,Ti,Hi,Wi,To,Ho,Wo,KT,KH,KW,st,sh,sw,pt,ph,pw,ut,uh,uw) or None
    for a plain linear layer.  `opts` (CallOpts): the operand type of A / W / out_bf16 and the caller's A/B presets for this launch."""
    a = IgemmArgs()
    if opts is not None:
        a.operand = 0 if fp8 else opts.operand          # (e4m3 operands: the 16-bit OUTPUT of such a launch is bfloat16)
        tile = tile or opts.igemm_tile
        debug_flags |= opts.igemm_debug_or
        a.disable_256, a.min_k_256, a.splitk_max_tiles = opts.igemm_disable_256, opts.igemm_min_k_256, opts.igemm_splitk_max_tiles
    a.A, a.A_lo, a.W, a.W_lo = ptr(A), ptr(A_lo), ptr(W), ptr(W_lo)
    a.bias, a.rowvec, a.residual, a.mul = ptr(bias), ptr(rowvec), ptr(residual), ptr(mul)
    a.out_f32, a.out_bf16, a.out_bf16_lo = ptr(out_f32), ptr(out_bf16), ptr(out_bf16_lo)
    a.a_batch_stride, a.w_batch_stride = a_batch_stride, w_batch_stride
    a.out_batch_stride, a.outb_batch_stride, a.res_batch_stride = out_batch_stride, outb_batch_stride, res_batch_stride
    a.w_tap_stride = w_tap_stride
    a.lda = lda if lda is not None else Cin
    a.ldw = ldw if ldw is not None else Cin
    if getattr(W, "_pd_fold", False):
        # packing.fold_weights: W = the `taps` slabs of W_hi followed by the `taps` slabs of W_lo -- two weight products per tap against one
        # activation gather (precision="fp16x2"); the caller describes the layer as it would for a single product
        if A_lo is not None or W_lo is not None or fp8:
            raise PrediffHipError("pd_igemm: folded (hi + lo) weights go with single-rounded 16-bit activations only")
        a.w_fold = taps
        w_tap_stride = w_tap_stride or N * a.ldw
        a.w_tap_stride = w_tap_stride
        taps = 2 * taps
    a.nbatch, a.M, a.N, a.Cin, a.taps = nbatch, M, N, Cin, taps
    if geom is None:
        geom = dict(B=1, Ti=1, Hi=1, Wi=M, To=1, Ho=1, Wo=M, KT=1, KH=1, KW=1, st=1, sh=1, sw=1, pt=0, ph=0, pw=0,
                    ut=1, uh=1, uw=1)
    for k, v in geom.items():
        setattr(a, k, v)
    a.rows_per_sample = rows_per_sample
    a.ld_rowvec = N if rowvec is None else rowvec.shape[-1]
    a.ld_res = ld_res if ld_res is not None else N
    a.res_period = res_period
    a.ld_mul = N
    a.act = ACT[act]
    a.ld_out = ld_out if ld_out is not None else N
    a.ld_outb = ld_outb if ld_outb is not None else N
    a.split = 1 if A_lo is not None else 0
    a.alpha = alpha
    a.tile = tile
    a.debug_flags = debug_flags
    a.fp8 = 1 if fp8 else 0           # A / W are e4m3 bytes (torch.float8_e4m3fn); tensor scales in alpha
    a.out_fp8_log2 = out_fp8_log2     # k > 0: out_bf16 is an e4m3 byte tensor receiving e4m3(v * 2^k)
    if splitk_ws is not None:       # fp32 workspace: lets the library split the K loop of small-grid, long-K launches
        a.splitk_ws, a.splitk_ws_elems = ptr(splitk_ws), splitk_ws.numel()
    _check(lib().pd_igemm(C.byref(a), stream_ptr()), "pd_igemm")


def conv_geom(B, in_thw, kernel, stride=(1, 1, 1), pad=(1, 1, 1), up=(1, 1, 1), out_thw=None, virt_thw=None):
    """virt_thw: size of the nearest-up-sampled input when it is not exactly in_thw*up (odd target sizes)."""
    Ti, Hi, Wi = in_thw
    KT, KH, KW = kernel
    if out_thw is None:
        v = virt_thw if virt_thw is not None else tuple(s * u for s, u in zip(in_thw, up))
        out_thw = tuple((s + 2 * p - k) // st + 1 for s, p, k, st in zip(v, pad, kernel, stride))
    To, Ho, Wo = out_thw
    vt, vh, vw = virt_thw if virt_thw is not None else (0, 0, 0)
    return dict(vT=vt, vH=vh, vW=vw, B=B, Ti=Ti, Hi=Hi, Wi=Wi, To=To, Ho=Ho, Wo=Wo, KT=KT, KH=KH, KW=KW, st=stride[0], sh=stride[1],
                sw=stride[2], pt=pad[0], ph=pad[1], pw=pad[2], ut=up[0], uh=up[1], uw=up[2])


def layernorm(x, gamma, beta, out, out_lo, rows, Cn, ld_out, eps=1e-5, opts=None):
    _check(lib().pd_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out), ptr(out_lo), rows, Cn, ld_out, eps, _opts_ref(opts), stream_ptr()),
           "pd_layernorm")


def layernorm_fp8(x, gamma, beta, out, rows, Cn, ld_out, fp8_scale, eps=1e-5):
    """LayerNorm -> e4m3 rows (value * fp8_scale, saturating): the A operand of an fp8 pd_igemm launch."""
    _check(lib().pd_layernorm_fp8(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, Cn, ld_out, eps, fp8_scale, stream_ptr()),
           "pd_layernorm_fp8")


def patch_merge_layernorm(x, gamma, beta, out, out_lo, B, T, H, W, Cn, ds, ld_out, eps=1e-5, pad_nearest=False, opts=None):
    _check(lib().pd_patch_merge_layernorm_ex(ptr(x), ptr(gamma), ptr(beta), ptr(out), ptr(out_lo), B, T, H, W, Cn,
                                             ds[0], ds[1], ds[2], ld_out, eps, 1 if pad_nearest else 0, _opts_ref(opts), stream_ptr()),
           "pd_patch_merge_layernorm")


def groupnorm_nchunk(S, Cn):
    return lib().pd_groupnorm_nchunk(S, Cn)


def groupnorm_silu(x, gamma, beta, partials, out, out_lo, B, S, Cn, G, ld_out, eps, silu=True, ss_scale=None,
                   ss_shift=None, ld_ss=0, opts=None):
    _check(lib().pd_groupnorm_silu(ptr(x), ptr(gamma), ptr(beta), ptr(ss_scale), ptr(ss_shift), ld_ss, ptr(partials),
                                   ptr(out), ptr(out_lo), B, S, Cn, G, ld_out, eps, 1 if silu else 0, _opts_ref(opts), stream_ptr()),
           "pd_groupnorm_silu")


def groupnorm_stats(x, partials, stats, B, S, Cn, G, eps):
    """stats (B, G, 2) fp32 = {mean, rstd} of GroupNorm(G, Cn, eps) over channels-last x (B, S, Cn)."""
    _check(lib().pd_groupnorm_stats(ptr(x), ptr(partials), ptr(stats), B, S, Cn, G, eps, stream_ptr()), "pd_groupnorm_stats")


def conv2d_gn_silu_supported(H, W, Cin, Cout, G):
    return bool(lib().pd_conv2d_gn_silu_supported(H, W, Cin, Cout, G))


def conv2d_gn_silu(x, stats, gamma, beta, W, bias, residual, out, N, H, Wd, Cin, Cout, G, opts=None):
    """GroupNorm -> SiLU -> Conv2d 3x3 (+ bias, + fp32 residual) in one launch (csrc/conv2d_gn.hip): the VAE ResBlock body."""
    _check(lib().pd_conv2d_gn_silu(ptr(x), ptr(stats), ptr(gamma), ptr(beta), ptr(W), ptr(bias), ptr(residual), ptr(out), N, H, Wd, Cin,
                                   Cout, G, _opts_ref(opts), stream_ptr()), "pd_conv2d_gn_silu")


def conv2d_up2(x, W, bias, out, N, H, Wd, Cin, Cout, opts=None):
    """nearest x2 -> Conv2d 3x3 pad 1 (+ bias) of Upsample2D in one launch; H, Wd = OUTPUT size (pd_conv2d_up2)."""
    _check(lib().pd_conv2d_up2(ptr(x), ptr(W), ptr(bias), ptr(out), N, H, Wd, Cin, Cout, _opts_ref(opts), stream_ptr()), "pd_conv2d_up2")


def groupnorm_silu_fp8(x, gamma, beta, partials, out, B, S, Cn, G, eps, fp8_scale, silu=True, ss_scale=None, ss_shift=None, ld_ss=0):
    _check(lib().pd_groupnorm_silu_fp8(ptr(x), ptr(gamma), ptr(beta), ptr(ss_scale), ptr(ss_shift), ld_ss, ptr(partials), ptr(out),
                                       B, S, Cn, G, eps, 1 if silu else 0, fp8_scale, stream_ptr()), "pd_groupnorm_silu_fp8")


def groupnorm_silu_bwd(x, dy, gamma, beta, fwd_partials, bwd_partials, dx, B, S, Cn, G, eps=1e-5, silu=True):
    _check(lib().pd_groupnorm_silu_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(fwd_partials), ptr(bwd_partials), ptr(dx), B, S, Cn, G,
                                       eps, 1 if silu else 0, stream_ptr()), "pd_groupnorm_silu_bwd")


def cast_rows(x, out, out_lo, n_samples, rows_in, row_off, rows_out, Cn, ld_in, ld_out, opts=None):
    _check(lib().pd_cast_rows(ptr(x), ptr(out), ptr(out_lo), n_samples, rows_in, row_off, rows_out, Cn, ld_in, ld_out,
                              _opts_ref(opts), stream_ptr()), "pd_cast_rows")


def cuboid_attention(*, qkv_bf16=None, qkv_f32=None, tok_index, bias, mask, out_bf16=None, out_bf16_lo=None,
                     out_f32=None, B, ntok, Cn, heads, nc, vol, ld_qkv, ld_out, scale, force_generic=False, out_fp8_log2=0, tok_out=None,
                     qkv_fp8_log2=0, opts=None):
    a = CuboidAttnArgs()
    a.operand = 0 if qkv_fp8_log2 else (opts.operand if opts is not None else 0)
    a.qkv_fp8_log2 = qkv_fp8_log2     # k > 0: qkv_bf16 is an e4m3 byte tensor holding q, k, v * 2^k; q k^T and attn v run on the fp8 MFMA
    a.qkv_bf16, a.qkv_f32, a.tok_index, a.bias, a.mask = ptr(qkv_bf16), ptr(qkv_f32), ptr(tok_index), ptr(bias), ptr(mask)
    a.out_bf16, a.out_bf16_lo, a.out_f32 = ptr(out_bf16), ptr(out_bf16_lo), ptr(out_f32)
    a.B, a.ntok, a.C, a.heads, a.nc, a.vol, a.ld_qkv, a.ld_out = B, ntok, Cn, heads, nc, vol, ld_qkv, ld_out
    a.scale = scale
    a.force_generic = 1 if force_generic else 0
    a.out_fp8_log2 = out_fp8_log2     # k > 0: out_bf16 is an e4m3 byte tensor receiving e4m3(o * 2^k) (MFMA cores only)
    a.tok_out = ptr(tok_out)          # [nc][vol] token that receives each slot's result (padding_type "nearest"), or None = tok_index
    _check(lib().pd_cuboid_attention(C.byref(a), stream_ptr()), "pd_cuboid_attention")


def cuboid_attention_bwd(*, qkv, d_out, tok_index, bias, mask, d_qkv, B, ntok, Cn, heads, nc, vol, ld_qkv, ld_dout, ld_dqkv, scale):
    _check(lib().pd_cuboid_attention_bwd(ptr(qkv), ptr(d_out), ptr(tok_index), ptr(bias), ptr(mask), ptr(d_qkv), B, ntok, Cn, heads, nc,
                                         vol, ld_qkv, ld_dout, ld_dqkv, scale, stream_ptr()), "pd_cuboid_attention_bwd")


def softmax_rows(x, out, out_lo, rows, n, ld_in, ld_out, opts=None):
    _check(lib().pd_softmax_rows(ptr(x), ptr(out), ptr(out_lo), rows, n, ld_in, ld_out, _opts_ref(opts), stream_ptr()), "pd_softmax_rows")


def unet_build_input(x, cond, out, B, T_in, T_out, HW, Cn, ld_out):
    _check(lib().pd_unet_build_input(ptr(x), ptr(cond), ptr(out), B, T_in, T_out, HW, Cn, ld_out, stream_ptr()),
           "pd_unet_build_input")


def timestep_freqs(dim, max_period=10000.0, device="cpu"):
    """exp(-ln(max_period) * arange(half) / half) in fp32, the reference's own expression (models/utils.py:79-81)."""
    import math
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device)


def timestep_embedding(t, freqs, out, B, dim):
    _check(lib().pd_timestep_embedding(ptr(t), ptr(freqs), ptr(out), B, dim, stream_ptr()), "pd_timestep_embedding")


def linear_small(x, W, b, out, M, K, N, act_in="none", act_out="none"):
    _check(lib().pd_linear_small(ptr(x), ptr(W), ptr(b), ptr(out), M, K, N, ACT[act_in], ACT[act_out], stream_ptr()),
           "pd_linear_small")


def add_rowtable(x, table, n_samples, rows, Cn):
    _check(lib().pd_add_rowtable(ptr(x), ptr(table), n_samples, rows, Cn, stream_ptr()), "pd_add_rowtable")


def add(a, b, out, n):
    _check(lib().pd_add(ptr(a), ptr(b), ptr(out), n, stream_ptr()), "pd_add")


def ddpm_step(zt, eps, noise, mean_shift, t, coef, T, out, B, per_sample, temperature=1.0, clip_denoised=False):
    _check(lib().pd_ddpm_step(ptr(zt), ptr(eps), ptr(noise), ptr(mean_shift), ptr(t), ptr(coef), T, ptr(out), B,
                              per_sample, temperature, 1 if clip_denoised else 0, stream_ptr()), "pd_ddpm_step")


def ddim_step(zt, eps, noise, coef, out, B, per_sample):
    _check(lib().pd_ddim_step(ptr(zt), ptr(eps), ptr(noise), ptr(coef), ptr(out), B, per_sample, stream_ptr()),
           "pd_ddim_step")


def nchw_to_nhwc(x, out, N, Cn, HW, ld_out):
    _check(lib().pd_nchw_to_nhwc(ptr(x), ptr(out), N, Cn, HW, ld_out, stream_ptr()), "pd_nchw_to_nhwc")


def nhwc_to_nchw(x, out, N, Cn, HW, ld_in):
    _check(lib().pd_nhwc_to_nchw(ptr(x), ptr(out), N, Cn, HW, ld_in, stream_ptr()), "pd_nhwc_to_nchw")


def sevir_skill_counts(pred, target, thresholds, divisor, counts, outer, T, inner, keep_seq):
    _check(lib().pd_sevir_skill_counts(ptr(pred), ptr(target), ptr(thresholds), thresholds.numel(), divisor, ptr(counts), outer, T,
                                       inner, 1 if keep_seq else 0, stream_ptr()), "pd_sevir_skill_counts")


def attn_block_fused_supported(Cn, heads, vol):
    return bool(lib().pd_attn_block_fused_supported(Cn, heads, vol))


def attn_block_fused(x, out, gamma, beta, Wqkv, bqkv, Wp, bp, tok_index, bias, mask, B, ntok, Cn, heads, nc, vol, scale, eps=1e-5,
                     tok_affine=None, opts=None):
    """tok_affine: (n_inner, outer, inner, slot) from cuboid_geometry.affine_form(tok_index), or None (the kernel loads the table)."""
    aff = (C.c_int32 * 4)(*tok_affine) if tok_affine is not None else None
    _check(lib().pd_attn_block_fused_ex(ptr(x), ptr(out), ptr(gamma), ptr(beta), ptr(Wqkv), ptr(bqkv), ptr(Wp), ptr(bp), ptr(tok_index),
                                        ptr(bias), ptr(mask), B, ntok, Cn, heads, nc, vol, scale, eps,
                                        C.cast(aff, C.c_void_p) if aff is not None else None, _opts_ref(opts), stream_ptr()),
           "pd_attn_block_fused")


def ffn_fused_supported(Cn, Hd):
    return bool(lib().pd_ffn_fused_supported(Cn, Hd))


def ffn_fused(x, out, gamma, beta, W1, b1, W2, b2, M, Cn, Hd, act="gelu", eps=1e-5, opts=None):
    _check(lib().pd_ffn_fused(ptr(x), ptr(out), ptr(gamma), ptr(beta), ptr(W1), ptr(b1), ptr(W2), ptr(b2), M, Cn, Hd, ACT[act], eps,
                              _opts_ref(opts), stream_ptr()), "pd_ffn_fused")


def attn_ffn_pair_supported(Cn, heads, hidden, vol, act="gelu"):
    return bool(lib().pd_attn_ffn_pair_supported(Cn, heads, hidden, vol, ACT[act]))


def attn_ffn_pair_cuboids_per_group(vol):
    return 2 if 1 <= vol <= 8 else 1              # == pd_attn_ffn_pair_cuboids_per_group (tests/test_host_logic.py compares)


def attn_ffn_pair(x, out, wstream, vecs, tok_index, B, ntok, nc, vol, scale, eps_attn=1e-5, eps_ffn=1e-5, tok_affine=None, units=256,
                  opts=None):
    """One (CuboidSelfAttentionLayer, PositionwiseFFN) pair of a block (units 256 or 512) in one launch (csrc/pair_block.hip).
    wstream / vecs: packing.pack_pair_block / pack_pair_vecs."""
    aff = (C.c_int32 * 4)(*tok_affine) if tok_affine is not None else None
    _check(lib().pd_attn_ffn_pair(ptr(x), ptr(out), ptr(wstream), ptr(vecs), ptr(tok_index),
                                  C.cast(aff, C.c_void_p) if aff is not None else None, B, ntok, nc, vol, units, scale, eps_attn,
                                  eps_ffn, _opts_ref(opts), stream_ptr()), "pd_attn_ffn_pair")


def attn_ffn_pair_split_ws_floats(B, ntok, units):
    return 2 * 4 * B * ntok * units               # == pd_attn_ffn_pair_split_ws_floats (tests/test_host_logic.py compares)


def attn_ffn_pair_split(x, out, wstream, wffn_split, vecs, tok_index, B, ntok, nc, vol, scale, ws, eps_attn=1e-5, eps_ffn=1e-5, tok_affine=None,
                        units=512, opts=None):
    """The (attention, FFN) pair at units 512 for small grids: two tile launches + two row sums, four workgroups per 64-row tile (csrc/pair_block.hip MODE 1 / 2
    + pair_split_sum_kernel).  wffn_split: packing.pack_pair_ffn_split; ws: fp32 workspace of attn_ffn_pair_split_ws_floats(...) elements."""
    aff = (C.c_int32 * 4)(*tok_affine) if tok_affine is not None else None
    _check(lib().pd_attn_ffn_pair_split(ptr(x), ptr(out), ptr(wstream), ptr(wffn_split), ptr(vecs), ptr(tok_index),
                                        C.cast(aff, C.c_void_p) if aff is not None else None, B, ntok, nc, vol, units, scale, eps_attn, eps_ffn,
                                        ptr(ws), ws.numel(), _opts_ref(opts), stream_ptr()), "pd_attn_ffn_pair_split")


def ffn_rows_supported(Cn, Hd, act="gelu"):
    return bool(lib().pd_ffn_rows_supported(Cn, Hd, ACT[act]))


def ffn_rows(x, out, wffn, vecs, rows, units, eps=1e-5, opts=None):
    """PositionwiseFFN alone on the pair kernel's FFN half (csrc/pair_block.hip MODE 2, one hidden slice): x (rows, units) fp32 -> out (may alias).
    wffn: packing.pack_pair_ffn_split(w1, w2, nsplit=1); vecs: packing.pack_pair_vecs (LayerNorm-2, b1, b2 are read)."""
    _check(lib().pd_ffn_rows(ptr(x), ptr(out), ptr(wffn), ptr(vecs), rows, units, eps, _opts_ref(opts), stream_ptr()), "pd_ffn_rows")
