import torch, sys, os
# pure streaming-write reference: how fast can 109 MB be written with torch fill / copy
x = torch.empty(53248, 1024, dtype=torch.bfloat16, device="cuda")
y = torch.randn(53248, 1024, device="cuda").to(torch.bfloat16)
for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy bf16->bf16", lambda: x.copy_(y))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name}: {us:.1f} us for {x.numel()*2/1e6:.0f} MB written -> {x.numel()*2/us/1e6:.2f} TB/s write")
