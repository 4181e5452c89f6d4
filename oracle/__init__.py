"""CPU oracle for the PreDiff sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) functional
restatement of the reference algorithm (gaozhihan/PreDiff) for the path named in
BASELINE.json:north_star: Earthformer-UNet denoiser, frame-wise KL-VAE and the
DDPM/DDIM sampling loop.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker -- never as
the thing measured or shipped.  The product (`prediff_amd/`) does not import it.

Parity pinning: the reference has no tests or golden vectors of its own (SURVEY.md
F2), so every function here is pinned against vectors captured by importing the
reference in the build container (tests/golden/gen_golden.py -> tests/golden/*.npz,
checked by tests/test_oracle_golden.py).  The DDIM update rule is NOT in the
reference: `oracle.diffusion.ddim_step` is "parity unpinned" beyond
the two schedule helpers the reference does ship.
"""
