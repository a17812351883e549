#!/usr/bin/env python3
"""Decode GEMV launch anatomy: per-launch time when the weights come from L2 (one matrix re-read), from the
256 MB Infinity Cache (a set that exceeds the 8 x 4 MB L2s but fits the memory-side cache) or from HBM
(1 GB set), plus the launch-gap floor of a trivial kernel.  hipGraph of `launches` launches, best of 5 replays."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mobilequant_amd as mq
from mobilequant_amd import ops
from mobilequant_amd._lib import MQ_F32

dev = torch.device("cuda:0")


def graph_time(fn, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


aq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); aq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)
oq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); oq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)

# launch-gap floor
mn, mx = ops.minmax_new(1, dev)
t = graph_time(lambda: [ops.minmax_new(1, dev) for _ in range(88)])
print(f"trivial kernel x88 in a graph: {t * 1e6 / 88:.2f} us per launch")

SETS = (("L2 (1 matrix)", 0), ("Infinity Cache (~100 MB set)", 100), ("HBM (~1 GB set)", 1000))
if "--hbm-only" in sys.argv:
    SETS = SETS[2:]
for K, N in ((2048, 2048), (2048, 2560), (5632, 2048), (2048, 11264)):
    mb = K * N / 1e6
    for label, total_mb in SETS:
        n = max(1, int(total_mb / mb))
        ws = []
        for i in range(n):
            w8 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
            colsum = w8.to(torch.int32).sum(1).to(torch.int32)
            wscale = torch.full((1,), 2e-4, device=dev); woff = torch.full((1,), 128.0, device=dev)
            alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, 128, wscale, woff, 128, colsum, K)
            ws.append((w8, alpha, wzp, ct))
        x = torch.randn(1, K, device=dev)
        out = torch.empty(1, N, device=dev)
        launches = max(44, n)

        def run():
            for i in range(launches):
                w8, alpha, wzp, ct = ws[i % n]
                ops.int8_linear_f32in(x, aq.scale, aq.offset, 0.0, 255.0, 128, w8, alpha, wzp, ct, None, out_scale=oq.scale,
                                      out_offset=oq.offset, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_F32, out=out)
        t = graph_time(run) / launches
        print(f"K={K:5d} N={N:5d} {mb:5.1f} MB  {label:30s} {t * 1e6:6.2f} us/launch  {mb / t / 1e6:6.2f} TB/s")
        del ws
        torch.cuda.empty_cache()
