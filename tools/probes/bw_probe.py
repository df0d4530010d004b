import torch, time
x=torch.empty(1<<30, dtype=torch.float32, device='cuda')  # 4 GiB
y=torch.empty_like(x)
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
ms=t(lambda: x.fill_(1.0)); print("fill 4GiB  %.3f ms  %.2f TB/s write"%(ms, 4.295/ms))
ms=t(lambda: y.copy_(x)); print("copy 4GiB  %.3f ms  %.2f TB/s (r+w)"%(ms, 8.59/ms))
ms=t(lambda: x.sum()); print("sum 4GiB  %.3f ms  %.2f TB/s read"%(ms, 4.295/ms))
