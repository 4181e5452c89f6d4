"""CPU (oracle only, no GPU): which weights carry the fp16 weight-rounding term of a denoiser forward -- the measurement behind
precision="fp16x2_lin" (DESIGN.md section 5).  Rounds one group of the v1 denoiser's matrices / filters to IEEE half at a time (everything else
exact, fp32 arithmetic) and reports the rel-L2 of the forward against the exact one.  usage: python tests/diag_weight_rounding_sweep.py   (a diagnostic of the test tree, not a pytest module)"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _templates as TP
from _cases import V1_UNET_CFG
from _weights import seeded_input, seeded_state_dict
from oracle import unet as OU
torch.set_num_threads(16)
sd = seeded_state_dict(TP.unet_template(V1_UNET_CFG, "v1_unet_schema.json"), 1234)
x = seeded_input("d50x", (1, 6, 16, 16, 64), 22); c = seeded_input("d50c", (1, 7, 16, 16, 64), 21); t = torch.tensor([999])
ref = OU.unet_forward(sd, V1_UNET_CFG, x, t, c)
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
keys2 = [k for k,v in sd.items() if torch.is_floating_point(v) and v.dim()>=2]
def group(k):
    if 'time_embed_blocks' in k or k.startswith('first_proj'):
        return 'conv3d' if ('in_layers.2' in k or 'out_layers.3' in k) else ('emb' if 'emb_layers' in k else 'resblock_other')
    if '.attn_l.' in k:
        return 'attn_qkv' if '.qkv.' in k else ('attn_proj' if '.proj.' in k else 'attn_other')
    if '.ffn_l.' in k:
        return 'ffn1' if 'ffn_1' in k else ('ffn2' if 'ffn_2' in k else 'ffn_other')
    return 'other'
groups = {}
for k in keys2: groups.setdefault(group(k), []).append(k)
print({g: len(v) for g,v in groups.items()})
def run(sel, dtype=torch.float16):
    s2 = dict(sd)
    for k in sel: s2[k] = sd[k].to(dtype).float()
    return rel(OU.unet_forward(s2, V1_UNET_CFG, x, t, c), ref)
print('all fp16', run(keys2))
for g, ks in groups.items():
    print(g, len(ks), 'fp16-rounded alone: %.3e' % run(ks), ' all-but-this: %.3e' % run([k for k in keys2 if k not in ks]))
# level split of conv3d
for lvl in ('0','1'):
    ks=[k for k in groups['conv3d'] if ('blocks.'+lvl+'.') in k]
    print('conv3d level',lvl,len(ks),'%.3e'%run(ks))
