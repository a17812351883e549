import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
H, S, D = 32, 2048, 64
q = torch.randn(1, H, S, D, device=dev); k = torch.randn(1, H, S, D, device=dev); v = torch.randn(1, H, S, D, device=dev)
p = torch.randn(1, H, S, S, device=dev)
print("qk bmm", t(lambda: torch.matmul(q, k.transpose(2, 3))), "us  (17.2 GFLOP, writes 537 MB)")
print("pv bmm", t(lambda: torch.matmul(p, v)), "us  (17.2 GFLOP, reads 537 MB)")
x = torch.randn(2048, 2048, device=dev)
for n in (2048, 256, 5632):
    w = torch.randn(n, 2048, device=dev)
    us = t(lambda: torch.nn.functional.linear(x, w)); print(f"linear 2048x2048 -> {n}: {us:.1f} us = {2*2048*2048*n/us/1e6:.1f} TFLOPS")
w = torch.randn(2048, 5632, device=dev); x2 = torch.randn(2048, 5632, device=dev)
us = t(lambda: torch.nn.functional.linear(x2, w)); print(f"linear 2048x5632 -> 2048: {us:.1f} us = {2*2048*2048*5632/us/1e6:.1f} TFLOPS")
