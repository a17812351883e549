import torch, sys, os
sys.path.insert(0, os.getcwd())
x = torch.empty(32, 2048, 2048, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
us = t(lambda: x.fill_(1.0)); print(f"fill 537 MB: {us:.1f} us = {x.numel()*4/us/1e6:.2f} TB/s")
y = torch.empty_like(x)
us = t(lambda: torch.add(x, 1.0, out=y)); print(f"read+write 2 x 537 MB: {us:.1f} us = {2*x.numel()*4/us/1e6:.2f} TB/s")
us = t(lambda: x.sum()); print(f"read 537 MB (sum): {us:.1f} us = {x.numel()*4/us/1e6:.2f} TB/s")
