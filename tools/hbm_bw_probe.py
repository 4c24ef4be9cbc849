import torch, time
x = torch.empty(2**29, dtype=torch.float32, device='cuda')  # 2 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms=t(lambda: x.fill_(1.0)); print(f"fill 2 GiB: {ms:.3f} ms {2**31/ms/1e6:.0f} GB/s")
ms=t(lambda: x.zero_()); print(f"zero 2 GiB: {ms:.3f} ms {2**31/ms/1e6:.0f} GB/s")
ms=t(lambda: y.copy_(x)); print(f"copy 2 GiB: {ms:.3f} ms {2*2**31/ms/1e6:.0f} GB/s (r+w)")
ms=t(lambda: x.sum()); print(f"sum 2 GiB: {ms:.3f} ms {2**31/ms/1e6:.0f} GB/s read")
