import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (R, C) in ((4096, 4096), (14336, 4096), (32768, 8192)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x))
    print(f"{R}x{C} bf16 torch copy_: {ms*1e3:.1f} us  {2*x.numel()*2/ms/1e6:.0f} GB/s")
    ms = t(lambda: torch.mul(x, 2, out=y))
    print(f"{R}x{C} bf16 torch mul: {ms*1e3:.1f} us  {2*x.numel()*2/ms/1e6:.0f} GB/s")
