#!/usr/bin/env python3
"""How many 8-bit output indices of the headline GEMM differ between the epilogue the kernels compute --
    index = sat_u8(rne(float(t) * (alpha / s_o) + (bias / s_o + o)))            (one fma on pre-divided constants) --
and the reference's quantizer expression on the same exact integer accumulator t --
    y = float(t) * alpha + bias;  index = clamp(rint(y / s_o) + o, 0, 255)       (qmodule.py:286-287, IEEE divide)?
Both evaluated in fp32 by torch on the GPU from torch._int_mm's exact int32 accumulators; the kernel's own output is compared with
form one (must be identical) and form two (the flip rate DESIGN.md 3 quotes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mobilequant_amd import ops
from mobilequant_amd._lib import MQ_U8
from bench_fr128 import to_tiled, dev

M, K, N = 2048, 2048, 5632
g = torch.Generator(device="cpu").manual_seed(3)
a = (torch.randn(M, K, generator=g) * 40).round().clamp(-128, 127).to(torch.int8).to(dev)
w = (torch.randn(N, K, generator=g) * 40).round().clamp(-128, 127).to(torch.int8).to(dev)
sa, sw = 0.031, (torch.rand(N, generator=g) * 4e-3 + 1e-3).to(dev)
alpha = (sa * sw).float()
bias = (torch.randn(N, generator=g) * 0.2).to(dev)
t = torch._int_mm(a, w.t().contiguous())                                   # exact int32 accumulators (zero points 0: t is the whole integer part)
y = t.float() * alpha + bias
lo, hi = torch.quantile(y.flatten()[:: 97].float(), 0.001), torch.quantile(y.flatten()[:: 97].float(), 0.999)
so = ((hi - lo) / 255).reshape(1)
oo = torch.round(-lo / so).reshape(1)
zero = torch.zeros(N, dtype=torch.int32, device=dev)
rs = a.to(torch.int32).sum(1, dtype=torch.int32)
got = ops.int8_linear(to_tiled(a), w, rs, alpha, zero, zero, bias, out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8, a_tiled_rows=M)
inv = (1.0 / so).float()
form1 = torch.clamp(torch.round(torch.addcmul(bias * inv + oo, t.float(), alpha * inv)), 0, 255)      # (addcmul is not fused: see below)
form1_fma = torch.clamp(torch.round(torch.fma(t.float(), alpha * inv, bias * inv + oo)) if hasattr(torch, "fma") else form1, 0, 255)
form2 = torch.clamp(torch.round(y / so) + oo, 0, 255)
gi = got.float()
n = gi.numel()
print(f"kernel vs one-fma form (torch, unfused multiply-add): {int((gi != form1).sum())} of {n} differ")
if hasattr(torch, "fma"):
    print(f"kernel vs one-fma form (torch.fma): {int((gi != form1_fma).sum())} of {n} differ")
d = (gi - form2).abs()
print(f"kernel vs the reference's divide form on exact accumulators: {int((d != 0).sum())} of {n} differ = {float((d != 0).float().mean()):.2e}, max |diff| {int(d.max())} LSB")
sat = float(((form2 == 0) | (form2 == 255)).float().mean())
print(f"(saturated outputs: {sat:.3%})")
