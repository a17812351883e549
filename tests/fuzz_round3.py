"""Random-shape fuzz of round 3's kernels (test infrastructure; run by hand on an MI355X: python tests/fuzz_round3.py [iterations]):
  * the 128-column generated GEMMs on fragment-blocked activations == the C++ tile kernels on the row-major image, bit for bit
    (residual epilogue with a 16-bit grid, both tile heights; segmented 8-bit index outputs);
  * mq_attention_quant at head_dim 64 / 128 / 256 against the oracle, single shot and fed in chunks through an image cache
    (chunked == single shot, bit for bit); at head_dim 64 the q rows prepared inside the attention workgroups == the prep kernel's image,
    and (round 5) the fp16 score contraction == the int8 one;
  * the LDS-staged image-only norm / quantize kernels (packed-convert bytes) == the generic kernels, random shapes, grids, signed and
    unsigned, RMSNorm / LayerNorm, occasional non-finite elements."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle import mq_oracle as O  # noqa: E402
from mobilequant_amd import ops  # noqa: E402
import mobilequant_amd._lib as L  # noqa: E402
import test_gpu_round2 as T2  # noqa: E402
import test_gpu_round3 as T3  # noqa: E402

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(3)
bad = 0
for it in range(iters):
    # ---- GEMMs ----------------------------------------------------------------------------------------------------------------------
    M = int(rng.choice([9, 16, 100, 129, 300, 1000, 2048, 2500]))
    N = 128 * int(rng.integers(1, 20))
    K = 256 * int(rng.integers(3, 12))
    zp0, bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    a_q, w_q, a_rs, alpha, w_zp, col_term, b = T3._gemm_operands(dev, M, N, K, 100 + it, zp0, bias)
    resid = torch.randn(M, N, device=dev)
    so, oo = torch.tensor([float(rng.uniform(1e-4, 1e-3))], device=dev), torch.tensor([float(rng.integers(20000, 45000))], device=dev)
    kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0)
    want = ops.int8_linear(a_q, w_q, None if zp0 else a_rs, alpha, w_zp, col_term, b, resid=resid, **kw)
    for tile in (128, 256):
        L.load().mq_gemm_set_residual_tile(tile)
        got = ops.int8_linear(T3._to_tiled(a_q), w_q, None if zp0 else a_rs, alpha, w_zp, col_term, b, resid=resid, a_tiled_rows=M, **kw)
        if not torch.equal(got, want):
            bad += 1
            print("RESIDUAL MISMATCH", M, N, K, tile, zp0, bias, float((got - want).abs().max()))
    L.load().mq_gemm_set_residual_tile(0)
    nseg = int(rng.integers(1, 4))
    cuts = sorted(set(int(c) * 4 for c in rng.integers(1, N // 4, size=nseg - 1))) + [N]
    grids = [(torch.tensor([float(rng.uniform(0.005, 0.05))], device=dev), torch.tensor([float(rng.integers(60, 200))], device=dev)) for _ in cuts]
    want = ops.int8_linear_segmented(a_q, w_q, a_rs, alpha * 0.02, w_zp, col_term, b, cuts, grids)
    got = ops.int8_linear_segmented(T3._to_tiled(a_q), w_q, a_rs, alpha * 0.02, w_zp, col_term, b, cuts, grids, a_tiled_rows=M)
    if M > 8 and not torch.equal(got, want):
        bad += 1
        print("SEGMENTED MISMATCH", M, N, K, cuts, int((got.int() - want.int()).abs().max()))
    # ---- attention --------------------------------------------------------------------------------------------------------------------
    D = int(rng.choice([64, 128, 256]))
    heads = int(rng.choice([1, 2, 4, 8]))
    kv = int(rng.choice([h for h in (1, 2, 4, 8) if heads % h == 0]))
    nch = int(rng.integers(1, 4))
    chunks = [64 * int(rng.integers(1, 4)) for _ in range(nch - 1)] + [int(rng.integers(2, 200))]
    S = sum(chunks)
    qkb, pvb = int(rng.choice([16, 16, 12, 0])), int(rng.choice([8, 8, 16, 0]))
    q, k, v, cos, sin, qk, pv = T3._case(S, heads, kv, D, D, seed=500 + it, qk_out_bits=qkb, pv_out_bits=pvb)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv, qk, pv)
    g = dict(qk_a=T2._grid_of(qk[0], dev), qk_b=T2._grid_of(qk[1], dev), qk_out=T2._grid_of(qk[2], dev), pv_a=T2._grid_of(pv[0], dev),
             pv_b=T2._grid_of(pv[1], dev), pv_out=T2._grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    whole = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv, g, head_dim=D)
    cache = ops.attention_image_cache(kv, D, S, dev)
    outs, p0 = [], 0
    for n in chunks:
        sl = slice(p0, p0 + n)
        outs.append(ops.attention_quant(t(q[sl]), t(k[sl]), t(v[sl]), t(cos[sl]), t(sin[sl]), heads, kv, g, head_dim=D, cache=cache, pos0=p0))
        p0 += n
    chunked = torch.cat(outs)
    got = whole.cpu().numpy()
    d = np.abs(got - want)
    span = float(want.max() - want.min())
    step = float(pv[2].scale) if pv[2] is not None else 0.0
    ok = np.isfinite(got).all() and d.max() <= max(1.001 * step, 2e-3 * span) and np.median(d) <= 2e-4 * span
    if not ok or not torch.equal(chunked, whole):
        bad += 1
        print("ATTENTION", "oracle" if not ok else "chunk", D, heads, kv, chunks, qkb, pvb, float(d.max()), step, float((chunked - whole).abs().max()))
    if D == 64:                                # q rows prepared inside the attention workgroups (default) == the prep kernel's q image
        L.load().mq_attention_set_fused_q(0)
        try:
            ref_q = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv, g, head_dim=D)
        finally:
            L.load().mq_attention_set_fused_q(1)
        if not torch.equal(ref_q, whole):
            bad += 1
            print("ATTENTION fused q", heads, kv, S, qkb, pvb, float((ref_q - whole).abs().max()))
        if qkb:                                # round 5: scores on fp16 MFMAs over the centred indices (default) == the int8 contraction
            L.load().mq_attention_set_f16(0)
            try:
                ref_i8 = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv, g, head_dim=D)
            finally:
                L.load().mq_attention_set_f16(1)
            if not torch.equal(ref_i8, whole):
                bad += 1
                print("ATTENTION f16 scores", heads, kv, S, qkb, pvb, float((ref_i8 - whole).abs().max()))
    # ---- image-only staged kernels (v_med3 / v_cvt_pk_u8_f32 / v_sad_u8 bytes) == the generic kernels -------------------------------------
    rows = int(rng.choice([64, 65, 100, 333, 1000, 2048]))
    cols = 64 * int(rng.integers(16, 65))
    signed = bool(rng.integers(0, 2))
    qmin, qmax, shift = (-128.0, 127.0, 0) if signed else (0.0, 255.0, 128)
    x = (torch.randn(rows, cols, device=dev) * float(rng.uniform(0.2, 30.0)))
    if rng.integers(0, 3) == 0:
        x[int(rng.integers(0, rows)), int(rng.integers(0, cols))] = float(rng.choice([np.nan, np.inf, -np.inf, 1e30]))
    sc = torch.tensor([float(rng.uniform(0.005, 0.2))], device=dev)
    of = torch.tensor([float(rng.integers(-100, 100) if signed else rng.integers(0, 256))], device=dev)
    unt = lambda qq, Mp: qq.view(Mp // 16, cols // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]   # noqa: E731
    if cols in (1024, 2048, 3072, 4096):
        L.load().mq_quantize_tiled_set_staged(0)
        try:
            q0, rs0 = ops.quantize_tiled(x, sc, of, qmin, qmax, shift)
        finally:
            L.load().mq_quantize_tiled_set_staged(1)
        q1, rs1 = ops.quantize_tiled(x, sc, of, qmin, qmax, shift)
        if not (torch.equal(rs0, rs1) and torch.equal(unt(q0, q0.shape[0]), unt(q1, q1.shape[0]))):
            bad += 1
            print("QUANTIZE staged", rows, cols, signed)
    ln = bool(rng.integers(0, 2))
    w = 1.0 + 0.2 * torch.randn(cols, device=dev)
    bb = 0.1 * torch.randn(cols, device=dev) if ln else None
    gin = (torch.tensor([float(rng.uniform(20, 400)) / 65535], device=dev), torch.tensor([32768.0], device=dev), 0.0, 65535.0) if rng.integers(0, 2) else None
    gout = (sc * 0.5, of, qmin, qmax)
    _, qr, rsr, sh, _ = ops.rmsnorm_quant(x, w, bb, 1e-5, gin, gout, emit_int8=True, layernorm=ln, emit_tiled=False, want_y=False, emit_rowmajor=True)
    _, _, rst, _, qt = ops.rmsnorm_quant(x, w, bb, 1e-5, gin, gout, emit_int8=True, layernorm=ln, emit_tiled=True, want_y=False, emit_rowmajor=False)
    if not (torch.equal(rsr, rst) and torch.equal(unt(qt, (rows + 15) // 16 * 16), qr.view(rows, cols))):
        bad += 1
        print("NORM staged", rows, cols, signed, ln, gin is not None)
    if it % 10 == 9:
        print(f"{it + 1} iterations, {bad} failures", flush=True)
torch.cuda.synchronize()
print("fuzz_round3:", "FAILED" if bad else "ok", f"({iters} iterations, {bad} failures)")
sys.exit(1 if bad else 0)
