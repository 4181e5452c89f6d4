"""CPU: host-side logic of the product (geometry tables, patterns, packing, C-ABI surface).  No GPU compute."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from _cases import ATTN_CASES, MASK_CASES, REORDER_CASES
from oracle import unet as OU
from prediff_amd import cuboid_geometry as G
from prediff_amd.patterns import CuboidSelfAttentionPatterns

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("i", range(len(REORDER_CASES)))
def test_tok_index_matches_reference_reorder(golden, i):
    shape, cuboid, strategy = REORDER_CASES[i]
    t = G.attention_tables(shape, cuboid, (0, 0, 0), strategy, "zeros")
    assert np.array_equal(t["tok_index"].numpy(), golden("cuboid_index")[f"reorder_{i}"])
    assert t["mask"] is None


@pytest.mark.parametrize("i", range(len(MASK_CASES)))
def test_mask_matches_reference(golden, i):
    shape, cuboid, shift, strategy, padding_type = MASK_CASES[i]
    g = golden("cuboid_index")
    t = G.attention_tables(shape, cuboid, shift, strategy, padding_type)
    if padding_type == "nearest":
        # the padded grid is a nearest-neighbour resize: every slot reads a token (none is empty), every token receives exactly one result
        T, H, W = shape
        assert t["tok_out"] is not None and t["affine"] is None and int(t["tok_index"].min()) >= 0
        recv = t["tok_out"].numpy().ravel()
        assert sorted(recv[recv >= 0].tolist()) == list(range(T * H * W))
    assert list(t["cuboid"]) + list(t["shift"]) == g[f"mask_{i}_clamped"].tolist()
    ref = g[f"mask_{i}"]
    if t["mask"] is None:
        assert ref.all()
    else:
        assert np.array_equal(t["mask"].numpy().astype(bool), ref)


@pytest.mark.parametrize("shape,cuboid,shift", [((22, 5, 3), (13, 4, 2), (0, 0, 0)), ((33, 4, 4), (13, 4, 4), (0, 0, 0)),
                                                ((26, 6, 7), (11, 4, 4), (3, 1, 2)), ((44, 3, 22), (13, 2, 13), (0, 0, 0))])
def test_nearest_padding_tables_follow_torch_interpolate(shape, cuboid, shift):
    """padding_type="nearest" on axes where torch's floor(dst * float32(in / out)) differs from the integer floor(dst * in / out)
    (22 -> 26, 33 -> 39, 44 -> 52, 26 -> 33): the gather table must equal pad (F.interpolate) -> roll -> reorder of the token-id grid,
    and the receiver table must invert reorder -> roll back -> un-pad (F.interpolate), exactly as models/utils.py:228-270 runs them."""
    strategy = ("l", "l", "l")
    t = G.attention_tables(shape, cuboid, shift, strategy, "nearest")
    T, H, W = shape
    ids = torch.arange(T * H * W, dtype=torch.float32).view(1, T, H, W, 1)
    x = OU._pad_thw(ids, t["pad"], "nearest")
    sh = t["shift"]
    x = torch.roll(x, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
    ref = OU.cuboid_reorder(x, t["cuboid"], strategy)[0, :, :, 0].long()
    assert torch.equal(t["tok_index"].long(), ref)
    # receivers: give every slot a unique value, undo the reorder / roll, un-pad, and see which slot each token got
    nc, vol = t["nc"], t["vol"]
    P = tuple(s + p for s, p in zip(shape, t["pad"]))
    slots = torch.arange(nc * vol, dtype=torch.float32).view(1, nc, vol, 1)
    y = OU.cuboid_reorder_reverse(slots, t["cuboid"], strategy, P)
    y = torch.roll(y, shifts=sh, dims=(1, 2, 3))
    got = OU._unpad_thw(y, t["pad"], "nearest").reshape(-1).long()          # token -> slot id whose result it receives
    tok_out = t["tok_out"].reshape(-1).long()
    assert torch.equal(tok_out[got], torch.arange(T * H * W))
    assert int((tok_out >= 0).sum()) == T * H * W


def test_nearest_source_index_is_not_the_integer_rule():
    """the case the round-4 advisor found: an axis of 22 padded to 26 -- on the way back token 11 receives padded position 12
    (floor(11 * float32(26 / 22)) = floor(12.99999)); integer arithmetic would say 13"""
    back = G.nearest_source_index(26, 22)
    assert back[11] == 12 and (11 * 26) // 22 == 13
    # the formula the patch-merge kernel (csrc/norm.hip) evaluates per element agrees with F.interpolate on every size it can see
    for n in range(1, 70):
        for Pn in range(n, n + 17):
            scale = np.float32(n) / np.float32(Pn)
            k = np.minimum(np.floor(np.arange(Pn, dtype=np.float32) * scale).astype(np.int64), n - 1)
            assert np.array_equal(k, G.nearest_source_index(n, Pn)), (n, Pn)


def test_tok_index_shift_and_pad_against_oracle_roll():
    """tok_index must equal pad -> roll(-shift) -> reorder of the token-id grid (with -1 in the padding)."""
    for shape, cuboid, shift, strategy in [((5, 7, 6), (2, 4, 4), (1, 2, 2), ("l", "l", "l")),
                                           ((5, 8, 8), (2, 4, 4), (1, 2, 2), ("l", "l", "l")),
                                           ((6, 8, 12), (3, 2, 4), (1, 1, 2), ("d", "l", "d"))]:
        t = G.attention_tables(shape, cuboid, shift, strategy, "zeros")
        T, H, W = shape
        ids = torch.arange(T * H * W, dtype=torch.float32).view(1, T, H, W, 1) + 1      # 0 = padding
        pad = t["pad"]
        x = torch.nn.functional.pad(ids, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
        sh = t["shift"]
        x = torch.roll(x, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
        ref = OU.cuboid_reorder(x, t["cuboid"], strategy)[0, :, :, 0].long() - 1
        assert torch.equal(t["tok_index"].long(), ref)


@pytest.mark.parametrize("i", range(len(ATTN_CASES)))
def test_relative_position_index(golden, i):
    assert np.array_equal(G.relative_position_index(ATTN_CASES[i]["cuboid"]).numpy(), golden("attn_layer")[f"relidx_{i}"])


def test_patterns_match_oracle():
    for name in CuboidSelfAttentionPatterns.list_keys():
        for shape in [(13, 16, 16, 256), (13, 8, 8, 512), (5, 8, 8, 64), (4, 4, 4, 8)]:
            a = CuboidSelfAttentionPatterns.get(name)(shape)
            b = OU.attention_pattern(name, shape)
            assert [list(map(tuple, x)) for x in a] == [list(map(tuple, x)) for x in b], (name, shape)
    with pytest.raises(KeyError):
        CuboidSelfAttentionPatterns.get("nope")


def test_packing_layout():
    from prediff_amd.packing import pack_conv, pack_linear
    w = torch.randn(5, 3, 3, 3, 3)
    hi, lo = pack_conv(w, True)
    assert hi.shape == (27, 5, 64) and hi.dtype == torch.bfloat16
    rec = (hi.float() + lo.float())[:, :, :3]
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                assert torch.allclose(rec[(kt * 3 + kh) * 3 + kw], w[:, :, kt, kh, kw], atol=1e-4)
    assert float(hi[:, :, 3:].float().abs().max()) == 0
    hi, lo = pack_linear(torch.randn(7, 100), False)
    assert hi.shape == (7, 128) and lo is None


def test_built_library_has_no_scratch_kernels():
    """Every kernel inside the built libprediff_hip.so -- all 236 of both operand builds, not only the pair kernel -- is free of scratch
    memory and VGPR spills (scripts/check_no_scratch.py reads the code objects bundled in the .so): a spill reload is a VMEM load in
    front of the counted weight-DMA waits of the fused kernels (attn_block_kernel<256, 2..4> spilled 4-12 registers)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_no_scratch", os.path.join(ROOT, "scripts", "check_no_scratch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from prediff_amd import _lib as L
    ks = m.kernels_of(L.LIB_PATH)
    assert len(ks) > 100
    bad = [(k[0], k[1], k[2]) for k in ks if k[1] > 0 or k[2] > 0]
    assert not bad, bad
    names = " ".join(k[0] for k in ks)
    for must in ("attn_block_kernelILi256ELi2", "attn_block_kernelILi256ELi3", "attn_block_kernelILi256ELi4", "pair_kernel", "igemm256_kernel"):
        assert must in names, must


def test_c_abi_exports_every_declared_symbol():
    """libprediff_hip.so loads and exports every function include/prediff_hip.h declares (no GPU needed)."""
    from prediff_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "prediff_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(pd_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert os.path.exists(L.LIB_PATH), "libprediff_hip.so missing: run __graft_entry__.build()"
    so = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} declared in prediff_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    # no exported DATA symbols: options travel per call (pd_call_opts / the args structs), nothing in the library is process-global
    assert not re.findall(r"^extern\s+[a-z ]+\*?\s*(pd_[a-z0-9_]+);", hdr, flags=re.M)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported_data = {ln.split()[-1] for ln in nm.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "BDd" and ln.split()[-1].startswith("pd_")}
    assert not exported_data, exported_data
    # the IEEE-half builds of the operand-typed entry points are internal (the public ones forward to them): present, not declared
    for name in ("pd_f16_igemm", "pd_f16_attn_ffn_pair", "pd_f16_layernorm", "pd_f16_groupnorm_silu", "pd_f16_cuboid_attention",
                 "pd_f16_ffn_fused", "pd_f16_attn_block_fused_ex", "pd_f16_conv2d_gn_silu", "pd_f16_cast_rows", "pd_f16_softmax_rows",
                 "pd_f16_patch_merge_layernorm_ex"):
        assert hasattr(so, name) and name not in declared, name
    assert L.lib().pd_abi_version() == L.ABI_VERSION == 4
    assert ctypes.sizeof(L.CallOpts) == so.pd_sizeof_call_opts()
    assert ctypes.sizeof(L.IgemmArgs) % 8 == 0


def test_no_gpu_means_loud_failure():
    from prediff_amd import _lib as L
    with pytest.raises(L.PrediffHipError):
        L._dev(torch.zeros(2))


def test_affine_form_of_token_tables():
    """cuboid_geometry.affine_form: the (n_inner, outer, inner, slot) strides reproduce the table entry for entry whenever they are
    returned (the fused attention block computes token ids from them instead of loading the table), and tables that are not affine
    -- a shifted window, a padded grid -- give None."""
    import numpy as np
    from prediff_amd.cuboid_geometry import attention_tables, affine_form
    for shape, cuboid in (((13, 16, 16), (13, 1, 1)), ((13, 16, 16), (1, 16, 1)), ((13, 16, 16), (1, 1, 16)), ((6, 8, 8), (6, 1, 1)),
                          ((25, 48, 48), (1, 48, 1)), ((4, 4, 4), (2, 2, 2))):
        tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
        tok = tabs["tok_index"].numpy().astype(np.int64)
        aff = tabs["affine"]
        if aff is None:
            continue
        n_inner, outer, inner, slot = aff
        c = np.arange(tok.shape[0])[:, None]
        s = np.arange(tok.shape[1])[None, :]
        assert np.array_equal((c // n_inner) * outer + (c % n_inner) * inner + s * slot, tok), (shape, cuboid)
    # the three axial patterns of the v1 level-0 grid ARE affine (the fast path is taken where it matters)
    for cuboid in ((13, 1, 1), (1, 16, 1), (1, 1, 16)):
        assert attention_tables((13, 16, 16), cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")["affine"] is not None
    # shifted (rolled) and padded tables are not
    assert attention_tables((4, 8, 8), (2, 4, 4), (1, 2, 2), ("l", "l", "l"), "zeros")["affine"] is None
    assert attention_tables((13, 16, 16), (5, 1, 1), (0, 0, 0), ("l", "l", "l"), "zeros")["affine"] is None
    import torch
    assert affine_form(torch.tensor([[0, 1], [2, -1]])) is None


def test_conv2d_gn_silu_supported_predicate():
    """Host-side geometry predicate of the fused VAE ResBlock kernel (no launch): the shapes of the v1 VAE's ResBlocks qualify, the
    32-channel stand-in VAE and odd grids do not (AutoencoderKL then runs pd_groupnorm_silu + pd_igemm)."""
    from prediff_amd import _lib as L
    assert L.conv2d_gn_silu_supported(128, 128, 128, 128, 32) and L.conv2d_gn_silu_supported(64, 64, 256, 256, 32)
    assert L.conv2d_gn_silu_supported(16, 16, 512, 512, 32) and L.conv2d_gn_silu_supported(32, 32, 256, 512, 32)
    assert not L.conv2d_gn_silu_supported(64, 64, 32, 64, 8)          # Cin % 64
    assert not L.conv2d_gn_silu_supported(12, 16, 128, 128, 32)        # H % 8
    assert not L.conv2d_gn_silu_supported(16, 24, 128, 128, 32)        # W % 16
    assert not L.conv2d_gn_silu_supported(16, 16, 128, 64, 32)         # Cout % 128
    assert not L.conv2d_gn_silu_supported(16, 16, 64, 128, 32)         # (Cin / G) % 4


def test_product_ddim_helpers_against_reference_golden(golden):
    """The PRODUCT's make_ddim_timesteps / make_ddim_sampling_parameters (prediff_amd/schedule.py) against what the imported
    reference recorded (tests/golden/gen_golden.py:246-254; reference diffusion/utils.py:42-70) -- not through the oracle."""
    from prediff_amd import schedule as S
    g = golden("schedule")
    ac = np.cumprod(1.0 - S.make_beta_schedule("linear", 1000)).astype(np.float32).astype(np.float64)
    for n in (10, 50, 100):
        steps = S.make_ddim_timesteps("uniform", n, 1000)
        assert steps.dtype == g[f"ddim_steps_{n}"].dtype and np.array_equal(steps, g[f"ddim_steps_{n}"])
        for eta in (0.0, 1.0):
            sig, a, ap = S.make_ddim_sampling_parameters(ac, np.minimum(steps, 999), eta)
            assert np.allclose(sig, g[f"ddim_sigma_{n}_{int(eta)}"], rtol=1e-12, atol=0)
            assert np.allclose(a, g[f"ddim_a_{n}"], rtol=1e-12, atol=0) and np.allclose(ap, g[f"ddim_aprev_{n}"], rtol=1e-12, atol=0)
    assert np.array_equal(S.make_ddim_timesteps("quad", 20, 1000), g["ddim_steps_quad_20"])
    with pytest.raises(NotImplementedError):
        S.make_ddim_timesteps("cubic", 10, 1000)


def test_pair_kernel_async_lds_check():
    """scripts/check_async_lds.py (also run by __graft_entry__.build()): the pair kernel's ISA keeps every register with an LDS read in
    flight untouched until the covering wait, and uses no scratch.  Needs hipcc (cross-compiles without a GPU)."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_async_lds.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 finding(s)" in out.stdout


@pytest.mark.parametrize("Cn", [256, 512])
def test_pair_packing_layout(Cn):
    """packing.pack_pair_block / pack_pair_vecs (the operands of pd_attn_ffn_pair, csrc/pair_block.hip): every weight element appears
    exactly once in the stream, at the fragment position the kernel's chunk enumeration expects (spot checks against the documented
    index formulas); the score table is the bias on each cuboid's diagonal block and -inf elsewhere; the group-packing rule equals the
    library's."""
    import torch
    from prediff_amd import _lib as L
    from prediff_amd.packing import pack_pair_block, pack_pair_vecs, pair_cuboids_per_group
    hid, HD, CT = 4 * Cn, Cn // 4, Cn // 16
    DT, CW = HD // 16, Cn // 256
    # integer-valued bf16-exact weights (<= 250: eight significant bits) that hash (matrix, row, column)
    enc = lambda m, R, K: ((torch.arange(R).reshape(-1, 1) * 7 + torch.arange(K).reshape(1, -1) * 3 + 41 * m) % 251).float()
    wqkv, wp, w1, w2 = enc(1, 3 * Cn, Cn), enc(2, Cn, Cn), enc(3, hid, Cn), enc(4, Cn, hid)
    ws = pack_pair_block(wqkv, wp, w1, w2).float()
    nchunks = 16 * CW * CW + 2 * (hid // 64) * CW
    assert tuple(ws.shape) == (nchunks, 32, 64, 8)
    total = sum(float(w.double().sum()) for w in (wqkv, wp, w1, w2))
    assert float(ws.double().sum()) == total
    elem = lambda W, F, KB, lane, j: float(W[16 * F + (lane & 15), 32 * KB + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)])
    # head h, kind (q, k, v), chunk s, fragment i: feature tile i % DT, k-step s * 32 / DT + i / DT
    for h, kind, s_, i, lane, j in ((0, 0, 0, 0, 0, 0), (3, 2, CW * CW - 1, 31, 63, 7), (1, 1, 0, 5, 17, 3)):
        c = h * 4 * CW * CW + kind * CW * CW + s_
        F = (kind * Cn + HD * h) // 16 + i % DT
        assert float(ws[c, i, lane, j]) == elem(wqkv, F, s_ * (32 // DT) + i // DT, lane, j)
    # proj slice of head h, chunk s: column tile i % CT, k-step HS h + s * 32 / CT + i / CT
    for h, s_, i, lane, j in ((0, 0, 0, 0, 0), (2, CW * CW - 1, 31, 40, 6)):
        c = h * 4 * CW * CW + 3 * CW * CW + s_
        assert float(ws[c, i, lane, j]) == elem(wp, i % CT, (HD // 32) * h + s_ * (32 // CT) + i // CT, lane, j)
    # hidden slices: W1_0, W1_1, (W2_j, W1_{j+2}) ...
    base = 16 * CW * CW
    assert float(ws[base, 6, 33, 2]) == elem(w1, 0 + (6 & 3), 6 >> 2, 33, 2)                                   # W1_0, chunk 0
    assert float(ws[base + CW, 9, 1, 5]) == elem(w1, 4 + (9 & 3), 9 >> 2, 1, 5)                                # W1_1, chunk 0
    assert float(ws[base + 2 * CW, 20, 7, 1]) == elem(w2, 20 % CT, 0 + 20 // CT, 7, 1)                         # W2_0, chunk 0
    assert float(ws[base + 3 * CW, 3, 50, 4]) == elem(w1, 8 + 3, 0, 50, 4)                                     # W1_2, chunk 0
    assert float(ws[nchunks - 1, 31, 63, 7]) == elem(w2, 31 % CT, 2 * (hid // 64 - 1) + (CW - 1) * (32 // CT) + 31 // CT, 63, 7)   # last chunk of W2_{n-1}
    assert L.attn_ffn_pair_split_ws_floats(3, 832, 512) == L.lib().pd_attn_ffn_pair_split_ws_floats(3, 832, 512)
    for vol in (1, 5, 8, 9, 13, 16):
        assert pair_cuboids_per_group(vol) == L.attn_ffn_pair_cuboids_per_group(vol) == L.lib().pd_attn_ffn_pair_cuboids_per_group(vol)
        bias = torch.randn(4, vol, vol)
        v = pack_pair_vecs(*(torch.full((Cn,), float(k)) for k in range(1, 7)), torch.full((hid,), 7.0), bias)
        assert v.numel() == 10 * Cn + 1024 and [float(v[k * Cn]) for k in range(6)] == [1, 2, 3, 4, 5, 6] and float(v[6 * Cn + hid - 1]) == 7
        rb = v[6 * Cn + hid:].reshape(4, 16, 16)
        n = pair_cuboids_per_group(vol)
        inside = torch.zeros(16, 16, dtype=torch.bool)
        for a in range(n):
            assert torch.equal(rb[:, a * vol:(a + 1) * vol, a * vol:(a + 1) * vol], bias)
            inside[a * vol:(a + 1) * vol, a * vol:(a + 1) * vol] = True
        assert bool(torch.isinf(rb[:, ~inside]).all()) and bool((rb[:, ~inside] < 0).all())


def test_product_cond_schedule_vs_reference_golden():
    """LatentDiffusion.make_cond_schedule (num_timesteps_cond > 1; reference latent_diffusion.py:295-299): cond_ids bit-equal to the
    reference's buffer (no GPU: the module constructs on CPU)."""
    import numpy as np
    import torch
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff_amd.latent_diffusion import LatentDiffusion
    from _cases import TINY_UNET_CFGS
    cfg = TINY_UNET_CFGS["axial"]
    net = CuboidTransformerUNet(**cfg, precision="fp32")
    for n_cond in (4, 1):
        ldm = LatentDiffusion(torch_nn_module=net, layout="NTHWC", data_shape=(2, 32, 32, 1), timesteps=1000, use_ema=False,
                              latent_shape=tuple(cfg["target_shape"]), first_stage_model=None, cond_stage_model=None,
                              num_timesteps_cond=n_cond)
        assert ldm.shorten_cond_schedule == (n_cond > 1)
        if n_cond > 1:
            g = np.load(os.path.join(ROOT, "tests", "golden", "cond_schedule.npz"))
            assert ldm.cond_ids.dtype == torch.long and np.array_equal(ldm.cond_ids.numpy(), g["cond_ids"])
            assert "cond_ids" in ldm.state_dict()           # a registered buffer, as in the reference


def test_scale_by_std_registers_the_reference_buffer():
    """scale_by_std=True keeps `scale_factor` as a persistent buffer (reference latent_diffusion.py:160-164), so a checkpoint written by the
    reference with that option strict-loads; without it `scale_factor` is a plain attribute and not in the state_dict."""
    import torch
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff_amd.latent_diffusion import LatentDiffusion
    from _cases import TINY_UNET_CFGS
    cfg = TINY_UNET_CFGS["axial"]
    net = CuboidTransformerUNet(**cfg, precision="fp32")
    kw = dict(torch_nn_module=net, layout="NTHWC", data_shape=(2, 32, 32, 1), timesteps=1000, use_ema=False,
              latent_shape=tuple(cfg["target_shape"]), first_stage_model=None, cond_stage_model=None)
    a = LatentDiffusion(**kw, scale_by_std=True, scale_factor=1.0)
    b = LatentDiffusion(**kw, scale_by_std=False, scale_factor=0.18215)
    assert "scale_factor" in a.state_dict() and a.state_dict()["scale_factor"].shape == ()
    assert "scale_factor" not in b.state_dict() and b.scale_factor == 0.18215
    sd = a.state_dict()
    sd["scale_factor"] = torch.tensor(0.25)                # what the reference stores after its first training batch
    a.load_state_dict(sd, strict=True)
    assert float(a.scale_factor) == 0.25
    z = torch.ones(2, 3)
    assert torch.equal(a.get_first_stage_encoding(z), 0.25 * z)


def test_modules_deepcopy_and_pickle():
    """A module that owns a pd_call_opts struct stays deepcopy- / pickle-able (EMA by deepcopy, DDP spawn, torch.save(module)):
    ctypes refuses structures with pointer members, so CallOpts serialises its integer members and drops the profiling pointer."""
    import copy
    import io
    import pickle
    import torch
    from prediff_amd import _lib as L
    from prediff_amd.autoencoder_kl import AutoencoderKL
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from _cases import TINY_UNET_CFGS, TINY_VAE_CFG
    o = L.CallOpts("fp16", pair_form=2, igemm_tile=3, trace=1234)
    for c in (copy.deepcopy(o), pickle.loads(pickle.dumps(o))):
        assert c._state() == o._state() and c.operand == 1 and c.igemm_tile == 3 and not c.trace
    assert o.replace(small_grid=1).small_grid == 1 and o.small_grid == 0 and o.replace().trace == 1234
    net = CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="fp16")
    vae = AutoencoderKL(**TINY_VAE_CFG, precision="bf16")
    for m in (net, vae):
        m2 = copy.deepcopy(m)
        assert m2.opts is not m.opts and m2.opts._state() == m.opts._state()
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        m3 = torch.load(buf, weights_only=False)
        assert m3.opts._state() == m.opts._state() and list(m3.state_dict()) == list(m.state_dict())
