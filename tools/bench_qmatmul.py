#!/usr/bin/env python3
"""mq_qmatmul against the simulated QMatMul path (HIP fake-quant kernels around the fp32 library bmm) at the attention block's shapes:
qk_bmm [32, S, 64] x [32, 64, S] -> 16-bit scores, pv_bmm [32, S, S] (16-bit probabilities) x [32, S, 64] -> 8-bit.  hipGraph of 10 calls."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mobilequant_amd as mq

dev = torch.device("cuda:0")


def timed(fn, n=10, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for S in (512, 2048):
    H, D = 32, 64
    q, k, v = (torch.randn(1, H, S, D, device=dev) for _ in range(3))
    p = torch.softmax(torch.randn(1, H, S, S, device=dev), -1)
    qk = mq.QMatMul(mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16))
    pv = mq.QMatMul(mq.QuantConfig(bitwidth=16), mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    sc = torch.matmul(q, k.transpose(2, 3))
    qk.set_scale_offset({"input": [float(q.min()), float(q.max())], "input2": [float(k.min()), float(k.max())], "output": [float(sc.min()), float(sc.max())]}, "buffer")
    o = torch.matmul(p, v)
    pv.set_scale_offset({"input": [0.0, float(p.max())], "input2": [float(v.min()), float(v.max())], "output": [float(o.min()), float(o.max())]}, "buffer")
    del sc, o
    kt = k.transpose(2, 3)
    with torch.no_grad():
        for name, mod, a, b, bytes_ in (("qk_bmm", qk, q, kt, 4 * (2 * H * S * D + H * S * S)), ("pv_bmm", pv, p, v, 4 * (H * S * S + 2 * H * S * D))):
            mod.int8_mode = "auto"
            t_int = timed(lambda: mod(a, b))
            mod.int8_mode = "off"
            t_sim = timed(lambda: mod(a, b))
            mod.int8_mode = "auto"
            print(f"S={S} {name}: mq_qmatmul {t_int:8.1f} us = {bytes_ / t_int / 1e6:6.2f} TB/s of algorithmic bytes ({bytes_ / 1e6:.0f} MB) | simulated path {t_sim:8.1f} us | x{t_sim / t_int:.1f}", flush=True)
