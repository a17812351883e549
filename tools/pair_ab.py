#!/usr/bin/env python3
"""A/B of mq_gemm_set_pair_mode on one box: 0 = one workgroup per tile and problem (512 workgroups), 1 = one workgroup per tile runs both
problems (256 workgroups).  Alternates the modes; hipGraph of 30 launches between HIP events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mobilequant_amd import ops, _lib
from mobilequant_amd._lib import MQ_U8
from bench_fr128 import operands, to_tiled, timed, dev

M, N, K = 2048, 5632, 2048
a_q, w_q, a_rs, alpha, w_zp, col_term = operands(M, N, K)
a_q = (torch.randn(M, K, device=dev) * 30).clamp(-128, 127).to(torch.int8)          # quantised-Gaussian operands, as in bench.py
w2 = (torch.randn(N, K, device=dev) * 30).clamp(-128, 127).to(torch.int8)
w1 = (torch.randn(N, K, device=dev) * 30).clamp(-128, 127).to(torch.int8)
a_t = to_tiled(a_q)
a_rs = a_q.to(torch.int32).sum(1, dtype=torch.int32)
h = [dict(w=w, alpha=alpha, w_zp=w_zp, col_term=col_term, bias=None, out_scale=torch.tensor([0.02], device=dev), out_offset=torch.tensor([128.0], device=dev)) for w in (w1, w2)]
lib = _lib.load()
for rep in range(4):
    for mode in (0, 1):
        lib.mq_gemm_set_pair_mode(mode)
        t = timed(lambda: ops.int8_linear_pair(a_t, M, a_rs, h[0], h[1], out_dtype=MQ_U8), n=30, reps=7)
        print(f"rep {rep} pair mode {mode}: {t:6.2f} us  ({2 * 2.0 * M * N * K / t / 1e6 / 5000:.3f} of peak)", flush=True)
lib.mq_gemm_set_pair_mode(0)
