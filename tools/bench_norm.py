#!/usr/bin/env python3
"""QRMSNorm.forward at prefill size: fused HIP kernel (mq_rmsnorm_quant) vs the composite torch ops around the HIP
quantizers, and the per-layer saving of the norm -> int8 -> q/k/v chain.  hipGraph of 20 forwards."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mobilequant_amd as mq
from mobilequant_amd.quantization.fp_ops import HFRMSNorm

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
for S, C in ((2048, 2048),):
    fp = HFRMSNorm(C, eps=1e-5).to(dev)
    norm = mq.QRMSNorm.from_float(fp, a16, a16, a8).requires_grad_(False)
    x = torch.randn(1, S, C, device=dev)
    norm.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
    lins = []
    for n in (2048, 256, 256):
        ql = mq.QLinear.from_float(torch.nn.Linear(C, n, bias=False).to(dev), a8, a8, a8).requires_grad_(False)
        ql.input_quantizer = None
        ql.set_scale_offset({"input": [-4.0, 4.0], "output": [-3.0, 3.0]}, "buffer")
        lins.append(ql)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm, self.q_proj, self.k_proj, self.v_proj = norm, *lins

        def forward(self, t):
            h = self.norm(t)
            return self.q_proj(h), self.k_proj(h), self.v_proj(h)
    ffn = []
    for n in (5632, 5632):
        ql = mq.QLinear.from_float(torch.nn.Linear(C, n, bias=False).to(dev), a8, a8, a8).requires_grad_(False)
        ql.input_quantizer = None
        ql.set_scale_offset({"input": [-4.0, 4.0], "output": [-3.0, 3.0]}, "buffer")
        ffn.append(ql)

    class FFN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm, self.w1, self.w3 = norm, *ffn

        def forward(self, t):
            h = self.norm(t)
            return self.w1(h), self.w3(h)
    blk = Block()
    fblk = FFN()
    mq.wire_integer_inputs(blk)
    mq.wire_integer_inputs(fblk)
    with torch.no_grad():
        for mode in ("auto", "off"):
            norm.fused_mode = mode
            tn = timeit(lambda: norm(x))
            tb = timeit(lambda: blk(x))
            tf = timeit(lambda: fblk(x))
            print(f"[{S}x{C}] fused_mode={mode:4s}  QRMSNorm.forward {tn:7.2f} us   norm + q/k/v block {tb:7.2f} us   norm + w1/w3 block {tf:7.2f} us")
