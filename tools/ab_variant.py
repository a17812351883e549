#!/usr/bin/env python3
"""Time GEMM variants of the headline shape across library builds: ab_variant.py <lib.so> [<lib.so> ...] -- v1 v2 ..."""
import os, sys, subprocess
args = sys.argv[1:]
split = args.index("--")
libs, variants = args[:split], [int(v) for v in args[split + 1:]]
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mobilequant_amd import ops, _lib
from mobilequant_amd._lib import MQ_U8
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K = 2048, 5632, 2048
a8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev); rs = a8.to(torch.int32).sum(1).to(torch.int32)
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev); colsum = w.to(torch.int32).sum(1).to(torch.int32)
one = torch.ones(1, device=dev)
alpha, wzp, ct = ops.linear_epilogue_prepare(one * 0.02, one * 131, 128, torch.rand(N, device=dev) * 1e-3 + 1e-4, torch.randint(0, 256, (N,), device=dev).float(), 128, colsum, K)
out = torch.empty(M, N, dtype=torch.uint8, device=dev); so, oo = one * 0.05, one * 128
fn = lambda: ops.int8_linear(a8, w, rs, alpha, wzp, ct, None, out_scale=so, out_offset=oo, out_qmin=0., out_qmax=255., out_dtype=MQ_U8, out=out)
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3
for v in [int(x) for x in sys.argv[1:]]:
    lib.mq_gemm_set_variant(v)
    print(f"  variant {v} {lib.mq_gemm_variant_name(v).decode():28s} {timeit(fn):7.2f} us")
'''
for rnd in range(2):
    for lib in libs:
        print(lib)
        env = dict(os.environ, MQ_LIB_PATH=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", code] + [str(v) for v in variants], env=env)
