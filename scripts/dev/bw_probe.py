"""What does a plain streaming read / copy reach on this device?  (context for the GroupNorm passes)"""
import torch
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (131, 524, 2096):
    x = torch.randn(mb * 1000 * 1000 // 2, device="cuda").half()
    y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x)); print(f"{mb} MB copy      : {ms*1e3:8.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s (read+write)")
    ms = t(lambda: x.sum(dtype=torch.float32)); print(f"{mb} MB sum       : {ms*1e3:8.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s (read)")
    ms = t(lambda: y.zero_()); print(f"{mb} MB memset    : {ms*1e3:8.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s (write)")
