#!/usr/bin/env python3
"""Packed-W4 generated kernel against the int8-image kernels on the same operands (hipGraph of 20 launches, HIP events)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mobilequant_amd import ops
from mobilequant_amd._lib import MQ_U8
from bench_fr128 import timed
dev = torch.device("cuda:0")
for M, N, K in ((2048, 5632, 2048), (2048, 16384, 2048), (2048, 2560, 2048), (2048, 2048, 5632), (4096, 5632, 2048)):
    g = torch.Generator().manual_seed(1)
    qw = torch.randint(0, 16, (N, K), generator=g, dtype=torch.uint8).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    sc, of = torch.tensor([8.0 / 255], device=dev), torch.tensor([128.0], device=dev)
    a_t, rs = ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128)
    packed = ops.pack_w4(qw)
    w8 = qw.view(torch.int8)
    colsum = qw.to(torch.int32).sum(1).to(torch.int32)
    wsc = torch.rand(N, generator=g).to(dev) * 1e-2 + 1e-3
    wof = torch.randint(0, 16, (N,), generator=g).float().to(dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(sc, of, 128, wsc, wof, 0, colsum, K)
    so, oo = torch.tensor([0.05], device=dev), torch.tensor([128.0], device=dev)
    out = torch.empty(M, N, dtype=torch.uint8, device=dev)
    import mobilequant_amd._lib as L
    if L.load().mq_gemm_set_w4_mode(0) == 0:                # per-wave unpack: experiment builds only (build.py --experiments)
        t4r = timed(lambda: ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, None, [(so, oo)], out=out))
    else:
        t4r = float("nan")
    L.load().mq_gemm_set_w4_mode(1)
    t4 = timed(lambda: ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, None, [(so, oo)], out=out))
    o4 = out.clone()
    t8 = timed(lambda: ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, None, out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=255.0,
                                       out_dtype=MQ_U8, out=out, a_tiled_rows=M))
    same = bool(torch.equal(o4, out))
    ops_ = 2.0 * M * N * K
    print(f"{M} x {K} -> {N}: packed W4, expanded per workgroup {t4:.2f} us ({ops_ / t4 / 1e6 / 5000:.3f} of peak) | per-wave unpack {t4r:.2f} us | int8 image {t8:.2f} us ({ops_ / t8 / 1e6 / 5000:.3f}) | identical {same}", flush=True)
