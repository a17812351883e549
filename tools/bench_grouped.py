#!/usr/bin/env python3
"""QLinear with per-group weight grids (group_size 128, 4-bit): the integer path (mq_w8a8_linear_grouped) against the simulated path
(HIP fake-quant + fp32 library GEMM) and against the per-channel integer path, at the headline shape.  hipGraph of 10 forwards."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mobilequant_amd as mq
from bench_fr128 import timed, dev

M, K, N = 2048, 2048, 5632
x = torch.randn(1, M, K, device=dev) * 1.3
for tag, wcfg in (("per-group g128 W4", mq.QuantConfig(bitwidth=4, is_per_channel=True, group_size=128)),
                  ("per-group g128 W8", mq.QuantConfig(bitwidth=8, is_per_channel=True, group_size=128)),
                  ("per-channel   W4", mq.QuantConfig(bitwidth=4, is_per_channel=True))):
    lin = torch.nn.Linear(K, N, bias=False).to(dev)
    q = mq.QLinear.from_float(lin, mq.QuantConfig(bitwidth=8), wcfg, mq.QuantConfig(bitwidth=8)).requires_grad_(False)
    q.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
    with torch.no_grad():
        q(x)
        t_int = timed(lambda: q(x), n=10)
        q.int8_mode = "off"
        q(x)
        t_sim = timed(lambda: q(x), n=10)
    print(f"{tag}: integer path {t_int:8.1f} us ({2.0 * M * N * K / t_int / 1e6 / 5000:.3f} of the int8 peak) | simulated path {t_sim:8.1f} us | x{t_sim / t_int:.1f}", flush=True)
