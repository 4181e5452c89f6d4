"""pd_groupnorm_silu (bf16 rows): the one-pass kernel against the statistics + apply pair of launches, v1 shapes (GPU box)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
DEV="cuda"
def timeit(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for B in (1,4,8,32):
  for S,Cn in ((3328,256),(832,512)):
    x=torch.randn(B,S,Cn,device=DEV); g=torch.ones(Cn,device=DEV); b=torch.zeros(Cn,device=DEV)
    part=torch.zeros(B*L.groupnorm_nchunk(S,Cn)*32*2,dtype=torch.float64,device=DEV)
    out=torch.empty(B*S,Cn,dtype=torch.bfloat16,device=DEV)
    r=[]
    for two in (1,0):
        o=L.CallOpts(groupnorm_two_launches=two)
        r.append(timeit(lambda: L.groupnorm_silu(x,g,b,part,out,None,B,S,Cn,32,Cn,1e-5,opts=o)))
    print(f"B={B} S={S} C={Cn}: two launches {r[0]:.1f} us, one pass {r[1]:.1f} us")
