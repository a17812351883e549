import torch, time
dev = torch.device("cuda:0")
M,K,N = 2048,2048,5632
a = torch.randint(-128,128,(M,K),dtype=torch.int8,device=dev)
w = torch.randint(-128,128,(N,K),dtype=torch.int8,device=dev)
def bench(fn, iters=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best=1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); e1.synchronize(); best=min(best,e0.elapsed_time(e1)/iters)
    return best*1e3
ops = 2.0*M*N*K
try:
    wt = w.t()
    t = bench(lambda: torch._int_mm(a, wt))
    print(f"torch._int_mm (hipBLASLt int8->int32): {t:.2f} us  {ops/t/1e6:.0f} TOPS")
except Exception as e:
    print("int_mm failed:", e)
ab = torch.randn(M,K,device=dev,dtype=torch.bfloat16); wb = torch.randn(N,K,device=dev,dtype=torch.bfloat16)
t = bench(lambda: torch.nn.functional.linear(ab, wb))
print(f"bf16 F.linear (hipBLASLt): {t:.2f} us  {ops/t/1e6:.0f} TFLOPS ({ops/t/1e6/2500*100:.0f}% of 2.5PF)")
af = torch.randn(M,K,device=dev); wf = torch.randn(N,K,device=dev)
t = bench(lambda: torch.nn.functional.linear(af, wf), 10)
print(f"fp32 F.linear: {t:.2f} us  {ops/t/1e6:.0f} TFLOPS")
try:
    a8 = torch.randn(M,K,device=dev).to(torch.float8_e4m3fn); w8 = torch.randn(N,K,device=dev).to(torch.float8_e4m3fn)
    sa = torch.tensor(1.0,device=dev); sb=torch.tensor(1.0,device=dev)
    t = bench(lambda: torch._scaled_mm(a8, w8.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16))
    print(f"fp8 _scaled_mm: {t:.2f} us {ops/t/1e6:.0f} TFLOPS")
except Exception as e:
    print("scaled_mm failed:", str(e)[:200])
