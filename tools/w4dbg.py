import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import mq_oracle as O
from test_gpu_round2 import T, exact_u8, tiled_image
from mobilequant_amd import ops
F32 = np.float32
dev = torch.device("cuda:0")
def run(M, N, K):
    rng = np.random.default_rng(M + N + K)
    qa = rng.integers(0, 256, size=(M, K)); qw = rng.integers(0, 16, size=(N, K))
    za = 100; zw = rng.integers(0, 16, size=N); sa = F32(0.02); sw = (rng.random(N, dtype=F32) * F32(1e-2) + F32(1e-3))
    a8 = (qa - 128).astype(np.int8)
    a_t = T(tiled_image(a8), dev); packed = T(O.pack_w4(qw, 0), dev)
    rs = T(a8.sum(1).astype(np.int32), dev); colsum = T(qw.sum(1).astype(np.int32), dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([sa], F32), dev), T(np.array([za], F32), dev), 128, T(sw, dev), T(zw.astype(F32), dev), 0, colsum, K)
    acc, pre = O.qlinear_int_exact(qa, za, sa, qw, zw, sw, None, blas=True)
    so = F32((np.percentile(pre, 99.5) - np.percentile(pre, 0.5)) / 255.0); oo = F32(np.rint(-np.percentile(pre, 0.5) / so))
    grids = [(torch.tensor([float(so)], device=dev), torch.tensor([float(oo)], device=dev))]
    want, _ = exact_u8(qa, za, sa, qw, zw, sw, None, so, oo)
    got = ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, None, grids).cpu().numpy()
    bad = got != want
    rows = np.unique(np.argwhere(bad)[:, 0]); cols = np.unique(np.argwhere(bad)[:, 1] % 176)
    print(f"M={M} N={N} K={K}: bad {int(bad.sum())}; rows {rows[:5]}..{rows[-5:] if len(rows) else ''} ({len(rows)}); cols mod 176: {cols[:40]}", flush=True)
for M, K in ((2048, 768), (2000, 2048), (2000, 768), (2040, 1024), (1990, 1024)):
    run(M, 5632, K)
