"""CPU: host-side logic of the product (geometry tables, patterns, packing, C-ABI surface).  No GPU compute."""
import ctypes
import os
import re

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
    if padding_type == "nearest":
        with pytest.raises(NotImplementedError):
            G.attention_tables(shape, cuboid, shift, strategy, padding_type)
        return
    t = G.attention_tables(shape, cuboid, shift, strategy, padding_type)
    assert list(t["cuboid"]) + list(t["shift"]) == g[f"mask_{i}_clamped"].tolist()
    ref = g[f"mask_{i}"]
    if t["mask"] is None:
        assert ref.all()
    else:
        assert np.array_equal(t["mask"].numpy().astype(bool), ref)


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


def test_c_abi_exports_every_declared_symbol():
    """libprediff_hip.so loads and exports every function include/prediff_hip.h declares (no GPU needed)."""
    from prediff_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "prediff_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(pd_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert os.path.exists(L.LIB_PATH), "libprediff_hip.so missing: run __graft_entry__.build()"
    so = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} declared in prediff_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert L.lib().pd_abi_version() == 1
    assert ctypes.sizeof(L.IgemmArgs) % 8 == 0


def test_no_gpu_means_loud_failure():
    from prediff_amd import _lib as L
    with pytest.raises(L.PrediffHipError):
        L._dev(torch.zeros(2))
