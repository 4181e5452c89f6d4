"""Stress check: the aligned DDPM loop (denoiser graphs on lane streams overlapped with the autograd guidance) must equal the
single-stream eager loop bit for bit in fp32, over repeated fresh graph captures.  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_configs as TC
from _weights import seeded_input, seeded_state_dict
from prediff_amd.presets import V1_ALIGN_ARGS
from prediff_amd import alignment as AL
ldm = TC._v1_ldm("fp32")
al = AL.SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid", model_args=dict(V1_ALIGN_ARGS))
al.model.load_state_dict(seeded_state_dict(al.model.state_dict(), 701))
al.model.cuda()
ldm.set_alignment(al.get_mean_shift)
B = 2
zt, zc = seeded_input("v1azt", (B, 6, 16, 16, 64), 12).cuda(), seeded_input("v1azc", (B, 7, 16, 16, 64), 13).cuda()
avg = torch.rand(B, 1).cuda()
tape = [zt] + [seeded_input(f"v1an{tt}", (B, 6, 16, 16, 64), 14).cuda() for tt in (99, 0)]
def run(graph, streams, steps=2):
    ldm.use_hip_graph, ldm.num_streams = graph, streams
    return ldm.sample(cond=zc, batch_size=B, timesteps=steps, use_alignment=True, alignment_kwargs={"avg_x_gt": avg}, return_decoded=False, noise_tape=tape)

for tt in (99, 0):
    t = torch.full((B,), tt, dtype=torch.long, device="cuda")
    noise = seeded_input(f"v1an{tt}", (B, 6, 16, 16, 64), 14).cuda()
    out = ldm.p_sample(zt=zt, zc=zc, t=t, y=None, use_alignment=True, alignment_kwargs={"avg_x_gt": avg}, noise=noise)

ref = run(False, 1)
bad = 0
for it in range(30):
    ldm._graphs.clear() if hasattr(ldm, "_graphs") else None
    a = run(True, 2, 2)
    b = run(False, 2, 2)
    ea, eb = torch.equal(a, ref), torch.equal(b, ref)
    if not (ea and eb):
        bad += 1
        print(f"iter {it}: lanes==ref {ea} ({float((a-ref).abs().max()):.3e}); eager==ref {eb} ({float((b-ref).abs().max()):.3e})", flush=True)
print("mismatching iterations:", bad, "of 30")
