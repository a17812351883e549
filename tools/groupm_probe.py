#!/usr/bin/env python3
"""Tile order of the N = 2048 residual GEMMs (o_proj, w2) against L2 fetch traffic: mq_gemm_set_group_m(g) makes each XCD's 32 tiles a
g x (32 / g) block of the 16 x 16 tile grid, so its private L2 fetches g A panels + 32 / g W panels.
    python tools/groupm_probe.py            # times every g (hipGraph of 20 launches, HIP events)
    python tools/groupm_probe.py 8          # 40 launches with g = 8 only (for a rocprofv3 --pmc FETCH_SIZE pass)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mobilequant_amd import ops
import mobilequant_amd._lib as L
from bench_fr128 import operands, to_tiled, timed, dev

SHAPES = (("o_proj", (2048, 2048, 2048)), ("w2", (2048, 2048, 5632)))


def setup(M, N, K):
    a_q, w_q, a_rs, alpha, w_zp, col_term = operands(M, N, K)
    a_t = to_tiled(a_q)
    resid, out = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
    so, oo = torch.tensor([3.1e-4], device=dev), torch.tensor([32768.0], device=dev)
    kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0, resid=resid, out=out, a_tiled_rows=M)
    return lambda: ops.int8_linear(a_t, w_q, a_rs, alpha, w_zp, col_term, None, **kw)


if __name__ == "__main__":
    lib = L.load()
    if len(sys.argv) > 1:
        g = int(sys.argv[1])
        lib.mq_gemm_set_group_m(g)
        for name, (M, N, K) in SHAPES:
            fn = setup(M, N, K)
            for _ in range(40):
                fn()
            torch.cuda.synchronize()
        sys.exit(0)
    for name, (M, N, K) in SHAPES:
        fn = setup(M, N, K)
        unique = M * K + N * K + M * N * 4
        for g in (1, 2, 4, 8, 16):
            lib.mq_gemm_set_group_m(g)
            bn = min(32 // g, N // 128)                # an XCD's 32 consecutive tiles: bm x bn of the 16 x 16 tile grid
            bm = 32 // bn
            panels = (bm + bn) * 128 * K * 8 + M * N * 4
            print(f"{name:7s} group_m {g:2d}: XCD block {bm:2d} x {bn:2d} tiles, model fetch {panels / 1e6:6.1f} MB (unique {unique / 1e6:5.1f} MB)  {timed(fn):6.2f} us", flush=True)
    lib.mq_gemm_set_group_m(0)
