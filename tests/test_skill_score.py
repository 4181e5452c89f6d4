"""SEVIRSkillScore: oracle vs reference golden on CPU (integer counts: bit exact), HIP kernel vs both on GPU."""
import numpy as np
import pytest
import torch

from _inputs import skill_inputs
from oracle import skill as OS

THR = (16, 74, 133, 160, 181, 219)


def test_oracle_counts_match_reference(golden):
    g = golden("skill_score")
    pred, target = skill_inputs()
    for mode, keep in (("0", False), ("1", True)):
        h1, m1, f1 = OS.counts(pred.numpy(), target.numpy(), 1, THR, keep)
        h2, m2, f2 = OS.counts(pred.flip(0).numpy(), target.numpy(), 1, THR, keep)
        assert np.array_equal(h1 + h2, g[f"hits_{mode}"].astype(np.int64))
        assert np.array_equal(m1 + m2, g[f"misses_{mode}"].astype(np.int64))
        assert np.array_equal(f1 + f2, g[f"fas_{mode}"].astype(np.int64))
        sc = OS.scores(h1 + h2, m1 + m2, f1 + f2)
        for i, thr in enumerate(THR):
            for met in ("csi", "pod", "sucr", "bias"):
                assert np.allclose(sc[met][i], g[f"score_{mode}_{thr}_{met}"], rtol=1e-6)
    # empty / all-NaN inputs count nothing
    z = np.full((1, 6, 4, 4, 1), np.nan, dtype=np.float32)
    assert all(int(v.sum()) == 0 for v in OS.counts(z, z, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_hip_skill_score_matches_reference(golden, mode):
    from prediff_amd.sevir_skill import SEVIRSkillScore
    g = golden("skill_score")
    pred, target = skill_inputs()
    m = SEVIRSkillScore(layout="NTHWC", mode=mode, seq_len=6, threshold_list=THR, metrics_list=("csi", "pod", "sucr", "bias"))
    m.update(pred.cuda(), target.cuda())
    m.update(pred.flip(0).cuda(), target.cuda())
    for name, st in (("hits", m.hits), ("misses", m.misses), ("fas", m.fas)):
        assert np.array_equal(st.cpu().numpy().astype(np.int64), g[f"{name}_{mode}"].astype(np.int64)), name      # bit exact
    res = m.compute()
    for thr in THR + ("avg",):
        for met in ("csi", "pod", "sucr", "bias"):
            assert np.allclose(np.asarray(res[thr][met], dtype=np.float64), g[f"score_{mode}_{thr}_{met}"], rtol=1e-6, atol=1e-9), (thr, met)
    m.reset()
    assert float(m.hits.sum()) == 0


@pytest.mark.gpu
def test_hip_skill_score_full_size_properties():
    """BASELINE size (32 members x 6 x 128 x 128): counts against the numpy oracle, plus hits+misses = #target>=T."""
    from prediff_amd.sevir_skill import SEVIRSkillScore
    g = torch.Generator().manual_seed(0)
    target = (torch.randint(0, 256, (32, 6, 128, 128, 1), generator=g).float() / 255) * (torch.rand(32, 6, 128, 128, 1, generator=g) > 0.6)
    pred = (target + 0.1 * torch.randn(target.shape, generator=g)).clamp(0, 1)
    m = SEVIRSkillScore(layout="NTHWC", mode="1", seq_len=6, threshold_list=THR)
    m.update(pred.cuda(), target.cuda())
    h, ms, fa = OS.counts(pred.numpy(), target.numpy(), 1, THR, True)
    assert np.array_equal(m.hits.cpu().numpy().astype(np.int64), h) and np.array_equal(m.misses.cpu().numpy().astype(np.int64), ms)
    assert np.array_equal(m.fas.cpu().numpy().astype(np.int64), fa)
    tv = (target.numpy().astype(np.float32) / np.float32(1 / 255.0))
    for i, thr in enumerate(THR):
        assert np.array_equal(h[i] + ms[i], (tv >= thr).sum(axis=(0, 2, 3, 4)))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["NHWT", "NTHW", "NHTW"])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_hip_skill_score_other_layouts(golden, layout, mode):
    """The reference's constructor default layout is "NHWT" (T last: neighbouring elements belong to different time steps) -- same
    frames permuted into it must give the golden counts of the NTHWC run, bit exact; also at the real frame size (128 x 128 x 6)."""
    from prediff_amd.sevir_skill import SEVIRSkillScore
    g = golden("skill_score")
    pred, target = skill_inputs()                       # N T H W C with C = 1
    perm = ["NTHW".index(c) for c in layout]
    p4, t4 = pred[..., 0].permute(*perm).contiguous(), target[..., 0].permute(*perm).contiguous()
    pf4 = pred.flip(0)[..., 0].permute(*perm).contiguous()
    m = SEVIRSkillScore(layout=layout, mode=mode, seq_len=6, threshold_list=THR)
    m.update(p4.cuda(), t4.cuda())
    m.update(pf4.cuda(), t4.cuda())
    for name, st in (("hits", m.hits), ("misses", m.misses), ("fas", m.fas)):
        assert np.array_equal(st.cpu().numpy().astype(np.int64), g[f"{name}_{mode}"].astype(np.int64)), name
    gen = torch.Generator().manual_seed(1)
    tgt = (torch.randint(0, 256, (4, 128, 128, 6), generator=gen).float() / 255) * (torch.rand(4, 128, 128, 6, generator=gen) > 0.6)
    prd = (tgt + 0.1 * torch.randn(tgt.shape, generator=gen)).clamp(0, 1)
    m2 = SEVIRSkillScore(mode="1", seq_len=6, threshold_list=THR)       # default layout NHWT
    m2.update(prd.cuda(), tgt.cuda())
    h, ms, fa = OS.counts(prd.numpy(), tgt.numpy(), 3, THR, True)
    assert np.array_equal(m2.hits.cpu().numpy().astype(np.int64), h) and np.array_equal(m2.misses.cpu().numpy().astype(np.int64), ms)
    assert np.array_equal(m2.fas.cpu().numpy().astype(np.int64), fa)
