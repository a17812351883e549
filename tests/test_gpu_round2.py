"""Round-2 GPU parity tests (through the C ABI): the generated-ISA GEMM kernels fed DIRECTLY with a numpy-built fragment-blocked
input against the exact integer oracle (every output), and the data-parallel calibration path under a real RCCL process group."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import mq_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    import mobilequant_amd._lib as L
    assert L.device_info()["arch"].startswith("gfx950")
    return torch.device("cuda:0")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def tiled_image(a8: np.ndarray) -> np.ndarray:
    """The documented fragment-blocked layout (include/mobilequant_amd.h, mq_quantize_tiled), built in numpy -- NOT by the
    kernel under test: 1-KiB blocks of 16 rows x 64 k ordered [row block][k block]; inside a block byte offset
    16 * ((row & 15) + 16 * ((k & 63) >> 4)) + (k & 15).  Rows past M are zero padding."""
    M, K = a8.shape
    Mp = (M + 15) // 16 * 16
    pad = np.zeros((Mp, K), np.int8)
    pad[:M] = a8
    # [rb, r, kb, kq, 16] -> [rb, kb, kq, r, 16]
    return np.ascontiguousarray(pad.reshape(Mp // 16, 16, K // 64, 4, 16).transpose(0, 2, 3, 1, 4)).reshape(Mp, K)


def _fma_f32_exact(t: int, a, b) -> float:
    """fl32(t * a + b) with ONE rounding (round to nearest even), by rational arithmetic -- the arbiter for the rare element
    where the float64 shortcut below double-rounds."""
    from fractions import Fraction
    fr = Fraction(int(t)) * Fraction(float(a)) + Fraction(float(b))
    if fr == 0:
        return 0.0
    sign, fr = (-1.0 if fr < 0 else 1.0), abs(fr)
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    if Fraction(2) ** e > fr:
        e -= 1
    scaled = fr / Fraction(2) ** (e - 23)
    q = scaled.numerator // scaled.denominator
    r = scaled - q
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (q & 1)):
        q += 1
    return sign * float(q) * 2.0 ** (e - 23)


def exact_u8(qa, za, sa, qw, zw, sw, bias, so, oo):
    """Exact 8-bit output indices of the integer path: clamp(rint(t * (alpha / so) + (bias / so + oo)), 0, 255) evaluated as the
    kernels do (fp32: one int -> float conversion, one fma) next to the exactly rounded quotient, so the test
    can state both bars: identical to the kernel formula, and within 1 LSB of exact quantisation of the exact value."""
    acc, _ = O.qlinear_int_exact(qa, za, sa, qw, zw, sw, None, blas=True)
    alpha = (F32(sa) * np.asarray(sw, F32)).astype(F32)
    inv = F32(1.0) / F32(so)
    a_p = (alpha * inv).astype(F32)
    b_p = ((np.asarray(bias, F32) * inv).astype(F32) + F32(oo)).astype(F32) if bias is not None else np.full(len(alpha), F32(oo), F32)
    t = acc.astype(F32)
    # fma in float64 then a single rounding to fp32 == fp32 fma whenever the product+sum is exactly representable in f64 (it is:
    # 24-bit x 24-bit product + 24-bit addend)
    v64 = t.astype(np.float64) * a_p.astype(np.float64) + b_p.astype(np.float64)
    v = v64.astype(F32)
    # candidates for double rounding: the float64 sum sits (almost) on a float32 rounding boundary AND on an index boundary
    frac = np.abs(v64 - np.rint(v64))
    for m, n in zip(*np.nonzero(np.abs(frac - 0.5) < 1e-6)):
        v[m, n] = _fma_f32_exact(int(acc[m, n]) if abs(int(acc[m, n])) < 2 ** 24 else int(t[m, n]), a_p[n], b_p[n])
    kernel = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    exact = np.clip(np.rint((acc.astype(np.float64) * alpha.astype(np.float64) + (0 if bias is None else np.asarray(bias, np.float64)))
                            / np.float64(so)) + np.float64(oo), 0, 255)
    return kernel, exact


@pytest.mark.parametrize("M,N,K,per_row,with_bias", [
    (2048, 5632, 2048, False, False),      # the headline shape (BASELINE.json configs[1])
    (2048, 5632, 2048, True, True),        # per-channel weights + bias (configs[2])
    (2000, 5632, 768, True, True),         # ragged M, the shortest K the free-running kernel serves (KT = 6)
    (1536, 5632, 1024, False, True),       # 6 x 32 = 192 tiles: the smallest grid
    (2048, 5632, 5632, True, False),       # long K (44 stages)
])
def test_generated_isa_gemm_kernels_vs_exact_oracle_every_output(dev, M, N, K, per_row, with_bias):
    """Variant 11 (free-running whole-kernel ISA, u8 / i8 outputs) and variant 9 (ping-pong ISA loop + C++ epilogue, every other
    output type) consume a fragment-blocked image built by numpy from the documented permutation; every output is compared with
    the exact integer oracle."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_F32, MQ_I8, MQ_U8
    assert ops.gemm_tiled_supported(M, N, K)
    rng = np.random.default_rng(M + N + K)
    qa = rng.integers(0, 256, size=(M, K))
    qw = rng.integers(0, 256, size=(N, K))
    za = int(rng.integers(0, 256))
    zw = rng.integers(0, 256, size=N) if per_row else np.full(N, int(rng.integers(0, 256)))
    sa = F32(0.02)
    sw = (rng.random(N, dtype=F32) * F32(1e-3) + F32(1e-4)) if per_row else np.full(N, F32(7e-4), F32)
    bias = rng.standard_normal(N, dtype=F32) if with_bias else None
    a8 = (qa - 128).astype(np.int8)
    a_t = T(tiled_image(a8), dev)
    w8 = T((qw - 128).astype(np.int8), dev)
    rs = T(a8.sum(1).astype(np.int32), dev)
    colsum = T((qw - 128).sum(1).astype(np.int32), dev)
    wsc = T(sw, dev) if per_row else T(sw[:1], dev)
    wof = T(zw.astype(F32), dev) if per_row else T(zw[:1].astype(F32), dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([sa], F32), dev), T(np.array([za], F32), dev), 128, wsc, wof, 128, colsum, K)
    b = T(bias, dev) if bias is not None else None
    # output grid sized to the data so that the clamp is exercised at both ends but most values are interior
    acc, pre = O.qlinear_int_exact(qa, za, sa, qw, zw, sw, bias, blas=True)
    so = F32((np.percentile(pre, 99.5) - np.percentile(pre, 0.5)) / 255.0)
    oo = F32(np.rint(-np.percentile(pre, 0.5) / so))
    one = torch.ones(1, device=dev)
    kw = dict(out_scale=one * float(so), out_offset=one * float(oo), out_qmin=0.0, out_qmax=255.0)
    kernel, exact = exact_u8(qa, za, sa, qw, zw, sw, bias, so, oo)
    lib = L.load()
    try:
        for variant in (11, 9):
            lib.mq_gemm_set_variant(variant)
            got = ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, b, out_dtype=MQ_U8, a_tiled_rows=M, **kw).cpu().numpy()
            assert got.shape == (M, N)
            assert np.array_equal(got, kernel), (variant, int((got != kernel).sum()))
            assert np.abs(got.astype(np.float64) - exact).max() <= 1 and (got == exact).mean() > 0.999
            got_i8 = ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, b, out_dtype=MQ_I8, a_tiled_rows=M, **kw).cpu().numpy()
            assert np.array_equal(got_i8.astype(np.int16) + 128, kernel.astype(np.int16)), variant
        lib.mq_gemm_set_variant(9)       # float output: the contraction itself, bit-exact (variant 9's C++ epilogue)
        got = ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, b, out_dtype=MQ_F32, a_tiled_rows=M).cpu().numpy()
        assert np.array_equal(bits(got), bits(pre))
    finally:
        lib.mq_gemm_set_variant(-1)
    # default dispatch: the 8-bit output takes the free-running kernel, and rows past a ragged M are never written
    canary = torch.full((M + 16, N), 77, dtype=torch.uint8, device=dev)
    ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, b, out_dtype=MQ_U8, a_tiled_rows=M, out=canary[:M], **kw)
    assert np.array_equal(canary[:M].cpu().numpy(), kernel) and bool((canary[M:] == 77).all())


def test_generated_isa_gemm_without_row_sums_and_with_zero_points_zero(dev):
    """a_rowsum = NULL (symmetric weights, w_zp == 0) through the free-running kernel."""
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    M, N, K = 2048, 5632, 1024
    rng = np.random.default_rng(3)
    qa = rng.integers(0, 256, size=(M, K))
    qw = rng.integers(-128, 128, size=(N, K))
    za, sa = 117, F32(0.03)
    sw = rng.random(N, dtype=F32) * F32(1e-3) + F32(1e-4)
    zw = np.zeros(N, np.int64)
    a8 = (qa - 128).astype(np.int8)
    colsum = T(qw.sum(1).astype(np.int32), dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([sa], F32), dev), T(np.array([za], F32), dev), 128, T(sw, dev),
                                                 T(zw.astype(F32), dev), 0, colsum, K)
    _, pre = O.qlinear_int_exact(qa, za, sa, qw, zw, sw, None, blas=True)
    so, oo = F32(np.abs(pre).max() / 120.0), F32(128.0)
    one = torch.ones(1, device=dev)
    got = ops.int8_linear(T(tiled_image(a8), dev), T(qw.astype(np.int8), dev), None, alpha, wzp, ct, None, out_dtype=MQ_U8,
                          a_tiled_rows=M, out_scale=one * float(so), out_offset=one * float(oo), out_qmin=0.0, out_qmax=255.0).cpu().numpy()
    kernel, _ = exact_u8(qa, za, sa, qw, zw, sw, None, so, oo)
    assert np.array_equal(got, kernel)


# ---- configs[4]: the collective path under a real RCCL group ----------------------------------------------------------
@pytest.fixture(scope="module")
def nccl_single_rank():
    import torch.distributed as dist
    if dist.is_initialized():
        yield
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("per_channel", [False, True])
def test_get_act_range_end_to_end_through_an_rccl_all_reduce(dev, nccl_single_rank, per_channel):
    """get_act_range on a llama-shaped model with the pack -> all_reduce(MAX, nccl = RCCL) -> unpack path FORCED in a 1-rank
    group (world == 1 normally returns early): the statistics must come back unchanged, exactly one collective is issued,
    and they equal torch's own min / max of every hooked tensor."""
    import torch.distributed as dist
    from mobilequant_amd import calibration as C
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    model = LlamaForCausalLM(LlamaShape.toy())
    model.reset_parameters(seed=11, std=0.3)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    samples = [torch.randint(0, 97, (1, 24), generator=g) for _ in range(5)]
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_reduce = counting
    try:
        forced = C.get_act_range(model, samples, per_channel=per_channel, force_collective=True)
    finally:
        dist.all_reduce = real
    assert calls["n"] == 1
    plain = C.get_act_range(model, samples, per_channel=per_channel)          # world == 1: no collective
    # reference statistics with torch ops in hooks (generate_act_range.py:55-69)
    want = {}

    def hook(name, matmul):
        def fn(m, xx, yy):
            items = [("input", xx[0]), ("output", yy[0] if isinstance(yy, tuple) else yy)] + ([("input2", xx[1])] if matmul else [])
            for field, t in items:
                t = t.detach()
                if per_channel:
                    t2 = t.reshape(-1, t.shape[-1])
                    cur = torch.stack((t2.min(0)[0], t2.max(0)[0]))
                    old = want.setdefault(name, {}).get(field)
                    want[name][field] = cur if old is None else torch.stack((torch.minimum(old[0], cur[0]), torch.maximum(old[1], cur[1])))
                else:
                    lo, hi = t.min().item(), t.max().item()
                    old = want.setdefault(name, {}).get(field)
                    want[name][field] = [lo, hi] if old is None else [min(old[0], lo), max(old[1], hi)]
        return fn
    hooks = [m.register_forward_hook(hook(n, C._is_matmul(m))) for n, m in model.named_modules() if C.is_calibrated_leaf(n, m)]
    with torch.no_grad():
        for s in samples:
            model(s.to(dev))
    for h in hooks:
        h.remove()
    assert forced.keys() == plain.keys() == want.keys() and len(forced) == 2 * 12 + 2
    for name in want:
        assert forced[name].keys() == want[name].keys()
        for field in want[name]:
            if per_channel:
                assert torch.equal(forced[name][field], plain[name][field]) and torch.equal(forced[name][field], want[name][field].cpu())
            else:
                assert forced[name][field] == plain[name][field] == want[name][field], (name, field)
