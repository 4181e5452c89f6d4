"""VAE encode / decode time at the v1 configuration, fused ResBlock (GroupNorm -> SiLU -> Conv2d in one launch) on / off.
Interleaved rounds in one process."""
import os, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd.autoencoder_kl import AutoencoderKL
from prediff_amd.presets import V1_VAE_CFG
from prediff_amd.seeding import seeded_state_dict
dev = torch.device("cuda")
vae = AutoencoderKL(**V1_VAE_CFG, precision="bf16")
vae.load_state_dict(seeded_state_dict(vae.state_dict(), 77))
vae = vae.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand(7 * T, 1, 128, 128, device=dev)
z = torch.randn(6 * T, 64, 16, 16, device=dev)
res = {}
if os.environ.get("VAE_FUSED_ONLY"):          # for rocprofv3: the production path alone, three passes each
    with torch.no_grad():
        for _ in range(3):
            vae.encode(x).mode(); vae.decode(z)
    torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    outs = {}
    for fused in (True, False):
        vae.fuse_resblock = fused
        outs[fused] = (vae.encode(x).mode().clone(), vae.decode(z).clone())
    e = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(f"fused vs un-fused: encode rel-L2 {e(outs[True][0], outs[False][0]):.2e}, decode {e(outs[True][1], outs[False][1]):.2e}")
    ts = {(f, n): [] for f in (True, False) for n in ("encode", "decode")}
    for r in range(5):
        for fused in (True, False):
            vae.fuse_resblock = fused
            for name, fn in (("encode", lambda: vae.encode(x).mode()), ("decode", lambda: vae.decode(z))):
                fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ts[fused, name].append((time.perf_counter() - t0) / 3 * 1e3)
for name, n, gf in (("encode", 7 * T, 68.0), ("decode", 6 * T, 155.2)):
    a, b = statistics.median(ts[True, name]), statistics.median(ts[False, name])
    print(f"{name} {n} frames: fused {a:.2f} ms ({n * gf / a:.0f} TFLOP/s), un-fused {b:.2f} ms ({n * gf / b:.0f} TFLOP/s)")
