"""igemm256_kernel: the two-phase K-tile (default) against the four-phase form of rounds 2-4 (debug_flags bit 64) on three Conv3d shapes:
the same products in the same order per accumulator, so the outputs are bit-equal (profiles/r05_g_two_phase_ab.txt).  Run on the GPU box."""
import torch, math, sys
sys.path.insert(0, "/root/repo")
from prediff_amd import _lib as L
from prediff_amd.packing import pack_conv
dev="cuda"
for (B,T,H,W,C) in [(2,13,16,16,256),(3,13,8,8,512),(1,5,7,9,64)]:
    g=torch.Generator(device="cpu").manual_seed(B+C)
    x=torch.randn(B,T,H,W,C,generator=g).to(dev); w=(torch.randn(C,C,3,3,3,generator=g)/math.sqrt(27*C)).to(dev)
    a=x.reshape(-1,C).to(torch.bfloat16).contiguous(); wp,_=pack_conv(w,False)
    M=B*T*H*W
    outs=[]
    for dbg in (0,64):
        o=torch.full((M,C),float("nan"),device=dev)
        L.igemm(a,wp,M=M,N=C,Cin=C,taps=27,w_tap_stride=C*C,geom=L.conv_geom(B,(T,H,W),(3,3,3)),out_f32=o,tile=7,debug_flags=dbg)
        outs.append(o)
    torch.cuda.synchronize()
    print((B,T,H,W,C),"bit-equal",torch.equal(outs[0],outs[1]), float(outs[1].abs().max()))
