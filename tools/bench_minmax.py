#!/usr/bin/env python3
"""HBM-bound statistics kernels: min/max per tensor (fresh and running statistic), per row (weights, a3) and per
column (per-channel activation calibration a12, absmax a13); fake-quant for reference.  hipGraph of 20 launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilequant_amd import ops

dev = torch.device("cuda:0")


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


for shape in ((2048, 2048), (5632, 2048), (2048, 5632), (16384, 2048)):
    x = torch.randn(*shape, device=dev)
    nbytes = x.numel() * 4
    mn, mx = ops.minmax_new(1, dev); ops.minmax_tensor_(x, mn, mx)
    rmn, rmx = ops.minmax_new(shape[0], dev)
    cmn, cmx = ops.minmax_new(shape[1], dev)
    y = torch.empty_like(x)
    sc, of = torch.full((1,), 0.03, device=dev), torch.full((1,), 128.0, device=dev)

    def fresh_atomics():
        a, b = ops.minmax_new(1, dev); ops.minmax_tensor_(x, a, b)
    cases = (("tensor fresh (init+atomics)", fresh_atomics), ("tensor fresh (partials+fold)", lambda: ops.minmax_tensor(x)), ("tensor running", lambda: ops.minmax_tensor_(x, mn, mx)),
             ("rows", lambda: ops.minmax_rows_(x, rmn, rmx)), ("cols", lambda: ops.minmax_cols_(x, cmn, cmx)),
             ("fake_quant (2x bytes)", lambda: ops.fake_quant(x, sc, of, 0, 255, out=y)))
    for name, fn in cases:
        t = timeit(fn)
        b = nbytes * (2 if name.startswith("fake") else 1)
        print(f"{str(shape):14s} {name:30s} {t:7.2f} us  {b / t / 1e6:5.2f} TB/s")
    assert mn.item() == x.min().item() and mx.item() == x.max().item()
    assert torch.equal(rmn, x.min(1).values) and torch.equal(cmx, x.max(0).values)
