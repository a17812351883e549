#!/usr/bin/env python3
"""A/B timing of the 128-column generated kernels against the C++ tile kernels they replace (hipGraph of 20 launches each)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mobilequant_amd import ops
import mobilequant_amd._lib as L

dev = torch.device("cuda:0")


def to_tiled(a_q):
    M, K = a_q.shape
    return a_q.view(M // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(M, K)


def timed(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1000 / n)
    return sorted(best)[len(best) // 2]


def operands(M, N, K):
    a_q = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w_q = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
    a_rs = a_q.to(torch.int32).sum(dim=1, dtype=torch.int32)
    alpha = torch.rand(N, device=dev) * 2e-4 + 1e-5
    w_zp = torch.zeros(N, dtype=torch.int32, device=dev)
    col_term = torch.randint(-50000, 50000, (N,), dtype=torch.int32, device=dev)
    return a_q, w_q, a_rs, alpha, w_zp, col_term


if __name__ == "__main__":
    for name, (M, N, K) in (("o_proj", (2048, 2048, 2048)), ("w2", (2048, 2048, 5632)), ("gemma w2", (2048, 2048, 16384)), ("M=4096 o", (4096, 2048, 2048))):
        a_q, w_q, a_rs, alpha, w_zp, col_term = operands(M, N, K)
        a_t = to_tiled(a_q)
        resid = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        so, oo = torch.tensor([3.1e-4], device=dev), torch.tensor([32768.0], device=dev)
        kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0, resid=resid, out=out)
        t_old = timed(lambda: ops.int8_linear(a_q, w_q, a_rs, alpha, w_zp, col_term, None, **kw))
        res = [f"rowmajor C++ {t_old:6.2f} us"]
        for tile in (128, 256, 512):
            if L.load().mq_gemm_set_residual_tile(tile) != 0:
                continue                                  # split-K: `python -m mobilequant_amd.build --experiments`
            t = timed(lambda: ops.int8_linear(a_t, w_q, a_rs, alpha, w_zp, col_term, None, a_tiled_rows=M, **kw))
            res.append(f"tiled {tile if tile < 512 else '256-row split-K'}{'-row' if tile < 512 else ''} {t:6.2f} us ({2.0 * M * N * K / t / 1e6:7.1f} TOPS)")
        L.load().mq_gemm_set_residual_tile(0)
        print(f"{name:10s} {M}x{N}x{K}: " + " | ".join(res), flush=True)

    M, K, ends = 2048, 2048, (2048, 2304, 2560)
    a_q, w_q, a_rs, alpha, w_zp, col_term = operands(M, ends[-1], K)
    a_t = to_tiled(a_q)
    grids = [(torch.tensor([0.011 * (i + 1)], device=dev), torch.tensor([100.0 + 20 * i], device=dev)) for i in range(3)]
    t_old = timed(lambda: ops.int8_linear_segmented(a_q, w_q, a_rs, alpha * 0.02, w_zp, col_term, None, ends, grids))
    L.load().mq_gemm_set_segmented_tile(128)
    t_256 = timed(lambda: ops.int8_linear_segmented(a_t, w_q, a_rs, alpha * 0.02, w_zp, col_term, None, ends, grids, a_tiled_rows=M))
    L.load().mq_gemm_set_segmented_tile(0)
    print(f"           256 x 128 tiles {t_256:6.2f} us")
    t_new = timed(lambda: ops.int8_linear_segmented(a_t, w_q, a_rs, alpha * 0.02, w_zp, col_term, None, ends, grids, a_tiled_rows=M))
    print(f"q|k|v      {M}x{ends[-1]}x{K}: rowmajor C++ {t_old:6.2f} us | tiled {t_new:6.2f} us ({2.0 * M * ends[-1] * K / t_new / 1e6:7.1f} TOPS)  (both allocate their output)")

    table = torch.randint(-128, 128, (65536,), dtype=torch.int8, device=dev)
    a = torch.randint(0, 256, (2048, 5632), dtype=torch.uint8, device=dev)
    b = torch.randint(0, 256, (2048, 5632), dtype=torch.uint8, device=dev)
    print(f"gated_lookup 2048x5632: rowmajor {timed(lambda: ops.gated_lookup(a, b, table)):6.2f} us | tiled {timed(lambda: ops.gated_lookup(a, b, table, tiled=True)):6.2f} us")

