#!/usr/bin/env python3
"""Soak test of the generated-ISA GEMM loop: many launches on changing inputs, every result compared with the C++
ping-pong path (bit-exact).  A phase-protocol bug would show up as a hang or a mismatch under timing jitter."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilequant_amd import ops
from mobilequant_amd._lib import MQ_I8, MQ_U8

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 2048, 5632, 2048
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
one = torch.ones(1, device=dev)
w8 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
colsum = w8.to(torch.int32).sum(1).to(torch.int32)
sc, of = one * 0.031, one * 121.0
alpha, wzp, ct = ops.linear_epilogue_prepare(sc, of, 128, torch.rand(N, device=dev) * 1e-3 + 1e-4, torch.randint(0, 256, (N,), device=dev).float(), 128, colsum, K)
kw = dict(out_scale=one * 0.05, out_offset=one * 128, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8)
bad = 0
for r in range(rounds):
    x = torch.randn(M, K, device=dev) * (1 + r % 5)
    q_rm, rs_rm = ops.quantize(x, sc, of, 0, 255, q_dtype=MQ_I8, shift=128, rows=M, want_row_sum=True)
    q_t, rs_t = ops.quantize_tiled(x, sc, of, 0, 255, 128)
    ref = ops.int8_linear(q_rm, w8, rs_rm, alpha, wzp, ct, None, **kw)
    outs = [ops.int8_linear(q_t, w8, rs_t, alpha, wzp, ct, None, a_tiled_rows=M, **kw) for _ in range(8)]   # back-to-back launches
    bad += sum(int(not torch.equal(ref, o)) for o in outs)
torch.cuda.synchronize()
print(f"soak: {rounds * 8} launches of the generated loop, mismatches: {bad}")
sys.exit(1 if bad else 0)
