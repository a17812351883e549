#!/usr/bin/env python3
"""Time mq_w8a8_linear / mq_w4a8_linear over the linear shapes of BASELINE.json's configs (M = 2048), every
tile variant or the built-in heuristic.  hipGraph of 20 launches between HIP events.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilequant_amd import ops, _lib
from mobilequant_amd._lib import MQ_U8, MQ_F32

dev = torch.device("cuda:0")
if "--lib" in sys.argv:                      # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
lib = _lib.load()
nvar = lib.mq_gemm_set_variant(-1)


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


def problem(M, N, K, w4):
    a8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    rs = a8.to(torch.int32).sum(1).to(torch.int32)
    if w4:
        nib = torch.randint(0, 16, (N, K), dtype=torch.uint8, device=dev)
        w = ops.pack_w4(nib); colsum = nib.to(torch.int32).sum(1).to(torch.int32); wshift = 0
    else:
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        colsum = w.to(torch.int32).sum(1).to(torch.int32); wshift = 128
    one = torch.ones(1, device=dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(one * 0.02, one * 131, 128, torch.rand(N, device=dev) * 1e-3 + 1e-4,
                                                 torch.randint(0, 16 if w4 else 256, (N,), device=dev).float(), wshift, colsum, K)
    out = torch.empty(M, N, dtype=torch.uint8, device=dev)
    so, oo = one * 0.05, one * 128
    return lambda: ops.int8_linear(a8, w, rs, alpha, wzp, ct, None, out_scale=so, out_offset=oo, out_qmin=0., out_qmax=255.,
                                   out_dtype=MQ_U8, w4=w4, out=out)


shapes = [("tinyllama q/o", 2048, 2048), ("tinyllama k/v", 256, 2048), ("tinyllama w1/w3", 5632, 2048), ("tinyllama w2", 2048, 5632),
          ("tinyllama q|k|v fused", 2560, 2048), ("gemma w1/w3", 16384, 2048), ("gemma w2", 2048, 16384)]
M = 2048
sweep = "--sweep" in sys.argv
print(f"{'shape':24s} {'N':>6s} {'K':>6s} {'wbits':>5s}  heuristic us / TOPS" + ("   | per-variant us" if sweep else ""))
for name, N, K in shapes:
    for w4 in (False, True):
        fn = problem(M, N, K, w4)
        lib.mq_gemm_set_variant(-1)
        t = timeit(fn)
        line = f"{name:24s} {N:6d} {K:6d} {'4' if w4 else '8':>5s}  {t:7.2f} us {2.0*M*N*K/t/1e6:6.0f} TOPS"
        if sweep:
            parts = []
            for v in range(nvar):
                if w4 and v == 7:
                    continue
                lib.mq_gemm_set_variant(v)
                parts.append(f"{lib.mq_gemm_variant_name(v).decode()}={timeit(fn):.1f}")
            lib.mq_gemm_set_variant(-1)
            line += "   | " + " ".join(parts)
        print(line, flush=True)
