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
    # the all-hooks path (what any graph gets): its statistics are EXACTLY torch's.  This package's own attention blocks otherwise take
    # qk_bmm.output / pv_bmm.input inside their fused score chain (round 5), whose probabilities sit a few ulp from torch's softmax
    # kernel -- tests/test_gpu_round5.py holds that path to the hook path within summation-order tolerance
    C.ActRangeCollector.fuse_attention_statistics = False
    C.ActRangeCollector.fuse_layer_statistics = False        # (round 6: the one-pass norms / gated product likewise -- tests/test_gpu_round6.py)
    try:
        forced = C.get_act_range(model, samples, per_channel=per_channel, force_collective=True)
        dist.all_reduce = real
        assert calls["n"] == 1
        plain = C.get_act_range(model, samples, per_channel=per_channel)          # world == 1: no collective
    finally:
        dist.all_reduce = real
        C.ActRangeCollector.fuse_attention_statistics = True
        C.ActRangeCollector.fuse_layer_statistics = True
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


# ---- a7: learnable weight clipping against the reference's frozen forward / autograd -------------------------------------------
def test_lwc_forward_backward_and_run_lwc_golden(dev):
    import mobilequant_amd as mq
    from conftest import load_meta, load_npz
    z = load_npz("lwc_cases.npz")
    for m in load_meta(z):
        t = m["id"]
        qz = mq.Quantizer(mq.QuantConfig(bitwidth=m["bitwidth"], is_symmetric=m["is_symmetric"], is_per_channel=m["is_per_channel"]))
        w = T(z[t + "_w"], dev).requires_grad_(True)
        qz.enable_lwc(w)
        assert qz.upbound_factor.shape == z[t + "_up"].shape and float(qz.upbound_factor.flatten()[0]) == 4.0     # init 4.0 (qmodule.py:135)
        with torch.no_grad():
            qz.upbound_factor.copy_(T(z[t + "_up"], dev))
            qz.lowbound_factor.copy_(T(z[t + "_lo"], dev))
        y = qz(w)
        (y * T(z[t + "_gy"], dev)).sum().backward()
        want_y = z[t + "_y"]
        d = np.abs(y.detach().cpu().numpy() - want_y)
        # the grid comes from sigmoid(factor) * range: one ulp of the device sigmoid moves values by ~1e-7 relative, never a grid step
        assert d.max() <= 4e-8 + 1e-6 * np.abs(want_y).max(), (t, d.max())
        assert np.allclose(qz.scale.detach().cpu().numpy().reshape(-1), z[t + "_scale"].reshape(-1), rtol=3e-7, atol=0)
        assert np.array_equal(qz.offset.detach().cpu().numpy().reshape(-1), z[t + "_offset"].reshape(-1))
        for got, key in ((qz.upbound_factor.grad, "_g_up"), (qz.lowbound_factor.grad, "_g_lo"), (w.grad, "_g_w")):
            want = z[t + key]
            assert got is not None and np.abs(got.cpu().numpy().reshape(-1) - want.reshape(-1)).max() <= 1e-5 * np.abs(want).max(), (t, key)
        with torch.no_grad():
            clamped = qz.run_lwc(T(z[t + "_w"], dev))
        assert np.abs(clamped.cpu().numpy() - z[t + "_clamped"]).max() <= 4e-8
        assert not qz.lwc and not hasattr(qz, "upbound_factor") and not hasattr(qz, "scale")      # run_lwc drops the state


# ---- a10: QMatMul against the reference at attention shapes ----------------------------------------------------------------------
def test_qmatmul_golden(dev):
    import mobilequant_amd as mq
    from conftest import load_meta, load_npz
    z = load_npz("qmatmul_cases.npz")
    for m in load_meta(z):
        t = m["id"]
        mod = mq.QMatMul(*(mq.QuantConfig(bitwidth=b) for b in m["bits"]))
        mod.set_scale_offset(m["act"], "buffer")
        a, b = T(z[t + "_a"], dev), T(z[t + "_b"], dev)
        if m["b_transposed"]:            # the reference passes k.transpose(2, 3): a non-contiguous view
            b = b.transpose(2, 3).contiguous().transpose(2, 3)
            assert not b.is_contiguous()
        with torch.no_grad():
            y = mod(a, b).cpu().numpy()
        lsb = float(mod.output_quantizer.scale)
        d = np.abs(y - z[t + "_y"])
        assert d.max() <= lsb * 1.001 and (d == 0).mean() > 0.99, (t, d.max() / lsb, (d == 0).mean())


# ---- perplexity proxy: NLL of the toy LM against the reference's W8A8-sim logits ----------------------------------------------
def test_toy_lm_nll_within_perplexity_bound_of_reference(dev):
    """north_star: "quantized perplexity within 0.05 of reference".  No checkpoints / datasets offline, so: the reference's own
    W8A8-sim logits of a 2-block toy LM on 96 tokens are frozen (tests/golden/toy_lm_nll.npz); the HIP path (integer linears,
    fused norms, chained int8 activations) must reproduce their NLL so that |perplexity - reference perplexity| <= 0.05."""
    import json
    import mobilequant_amd as mq
    from conftest import load_npz
    from toy_models import ToyLM, apply_mixed_precision
    z = load_npz("toy_lm_nll.npz")
    m = ToyLM().eval()
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd|")})
    m = m.to(dev)
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(m, a8, a8)
    apply_mixed_precision(m, mq)
    mq.set_scale_and_offset(m, json.loads(str(z["act"])), "buffer")
    mq.wire_integer_inputs(m)
    labels = torch.from_numpy(z["labels"]).to(dev)
    nll = lambda lg: float(torch.nn.functional.cross_entropy(lg[0], labels))       # noqa: E731
    with torch.no_grad():
        for mode in ("auto", "off"):            # integer path and simulated (HIP fake-quant) path
            for mod in m.modules():
                if isinstance(mod, mq.QLinear):
                    mod.int8_mode = mode
            got = nll(m(T(z["x"], dev)))
            ref = float(z["nll_w8a8"])
            assert abs(np.exp(got) - np.exp(ref)) <= 0.05, (mode, got, ref, np.exp(got) - np.exp(ref))
            assert abs(got - ref) <= 1e-3, (mode, got, ref)
    assert abs(float(z["nll_w8a8"]) - float(z["nll_fp"])) < 0.05           # the fixture itself: quantisation moved the NLL, mildly


# ---- n1 / f3 / f4: SmoothQuant statistics, fold, LET temporaries, run-time channel scale ------------------------------------
def test_chan_scale_fused_quantize_bit_exact_vs_reference(dev):
    """mq_quantize / mq_quantize_tiled with chan_scale: the index of x / s, bit-exact against the reference Quantizer applied to
    x / s (scales from smoothquant.py:60-62); same row sums in both layouts; NULL chan_scale is the plain kernel."""
    from conftest import load_npz
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_I8, MQ_U8
    z = load_npz("smooth_cases.npz")
    x, s = T(z["cs_x"], dev), T(z["cs_scales"], dev)
    lo, hi = (float(v) for v in z["cs_range"])
    sc, of, qmin, qmax = O.scale_offset_from_min_max(lo, hi, 8, False)
    sct, oft = T(np.array([sc], F32), dev), T(np.array([of], F32), dev)
    q = ops.quantize(x, sct, oft, qmin, qmax, q_dtype=MQ_U8, rows=x.shape[0], chan_scale=s)
    assert np.array_equal(q.cpu().numpy().astype(np.float32), z["cs_index"])
    q8, rs = ops.quantize(x, sct, oft, qmin, qmax, q_dtype=MQ_I8, shift=128, rows=x.shape[0], want_row_sum=True, chan_scale=s)
    assert np.array_equal(q8.cpu().numpy().astype(np.int32) + 128, z["cs_index"].astype(np.int32))
    assert np.array_equal(rs.cpu().numpy(), (z["cs_index"].astype(np.int64) - 128).sum(1))
    qt, rst = ops.quantize_tiled(x, sct, oft, qmin, qmax, 128, chan_scale=s)
    M, K = x.shape
    Mp = (M + 15) // 16 * 16
    back = qt.view(Mp // 16, K // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, K)[:M]
    assert torch.equal(back, q8) and torch.equal(rst, rs)
    # wide rows take the 16-elements-per-lane kernel: compare it with the generic one through an unaligned view
    g = torch.Generator().manual_seed(1)
    xb = torch.randn(64, 1024, generator=g).to(dev) * 3
    sb = (torch.rand(1024, generator=g) + 0.5).to(dev)
    a = ops.quantize(xb, sct, oft, qmin, qmax, q_dtype=MQ_U8, rows=64, chan_scale=sb)
    want = O.quantize_index((xb.cpu().numpy() / sb.cpu().numpy()).astype(F32), sc, of, qmin, qmax)
    assert np.array_equal(a.cpu().numpy().astype(F32), want)
    assert torch.equal(ops.quantize(xb, sct, oft, qmin, qmax, q_dtype=MQ_U8, rows=64, chan_scale=None),
                       ops.quantize(xb, sct, oft, qmin, qmax, q_dtype=MQ_U8, rows=64))


def test_qlinear_run_time_channel_scale_equals_offline_fold(dev):
    """QLinear.set_input_channel_scale(s): out = Qout(linear(Qin(x / s), Qw(W * s))) on the integer path == the same module with
    the scale folded by hand (x / s fed in, weight W * s), and within one output LSB of the simulated path."""
    import mobilequant_amd as mq
    from conftest import load_npz
    z = load_npz("smooth_cases.npz")
    x, s, w = T(z["cs_x"], dev), T(z["cs_scales"], dev), T(z["cs_w"], dev)
    a8 = mq.QuantConfig(bitwidth=8)
    lo, hi = (float(v) for v in z["cs_range"])

    def make(weight):
        lin = torch.nn.Linear(128, 96, bias=False).to(dev)
        with torch.no_grad():
            lin.weight.copy_(weight)
        ql = mq.QLinear.from_float(lin, a8, mq.QuantConfig(bitwidth=8, is_per_channel=True), a8).requires_grad_(False)
        y = (x / s) @ (w * s).T
        ql.set_scale_offset({"input": [lo, hi], "output": [float(y.min()), float(y.max())]}, "buffer")
        return ql
    run_time, folded = make(w).set_input_channel_scale(s), make(w * s.view(1, -1))
    assert np.array_equal((w * s.view(1, -1)).cpu().numpy(), z["cs_w_scaled"])
    with torch.no_grad():
        assert run_time._int8_ready(x, run_time._effective_weight(run_time.weight))
        y_rt = run_time(x)
        y_fold = folded(x / s)
        assert torch.equal(y_rt, y_fold)
        run_time.int8_mode = "off"
        y_sim = run_time(x)
    lsb = float(run_time.output_quantizer.scale)
    d = (y_rt - y_sim).abs()
    assert float(d.max()) <= lsb * 1.001 and float((d == 0).float().mean()) > 0.99


def test_smoothquant_statistics_and_fold_vs_reference(dev):
    """get_act_scales (device-resident absmax statistics, generate_act_scale_shift.py:42-93) and smooth_lm (smoothquant.py:109-139)
    on the llama-shaped model against the reference's real model: the statistic itself exact (first hooked tensor), deeper ones
    within the fp32 summation-order noise of the model's own GEMMs; folded weights within 3e-6 relative (torch's device pow vs CPU
    pow); the fold is function preserving."""
    from conftest import load_npz
    from test_llama_host import llama_from_fixture
    from mobilequant_amd import smoothquant as S
    z = load_npz("smooth_cases.npz")
    for tag in ("gqa", "mha"):
        m = llama_from_fixture(z, tag).to(dev)
        ids = [torch.from_numpy(r[None]).long() for r in z[tag + "_ids"]]
        scales = S.get_act_scales(m, ids)
        want = {k.split("|", 2)[2]: z[k] for k in z.files if k.startswith(f"{tag}|scale|")}
        ours = {("model." + k if not k.startswith("lm_head") else k): v for k, v in scales.items()}
        assert ours.keys() == want.keys(), sorted(set(ours) ^ set(want))[:5]
        first = "model.layers.0.input_layernorm_input"          # embeddings: nothing but the statistic kernel in front -> exact
        assert np.array_equal(ours[first].numpy(), want[first])
        for k in want:     # deeper tensors sit behind fp32 library GEMMs / softmax whose summation order differs from the CPU's
            assert np.allclose(ours[k].numpy(), want[k], rtol=2e-5, atol=1e-6), k
        with torch.no_grad():
            before = m(ids[0].to(dev))
        S.smooth_lm(m, {k[len("model."):] if k.startswith("model.") else k: torch.from_numpy(v) for k, v in want.items()}, 0.5)
        for k in z.files:
            if k.startswith(f"{tag}|smoothed|") and "rotary" not in k:
                name = k.split("|", 2)[2]
                name = name[len("model."):] if name.startswith("model.") else name
                got = m.state_dict()[name].cpu().numpy()
                assert np.allclose(got, z[k], rtol=3e-6, atol=1e-8), name
        with torch.no_grad():
            after = m(ids[0].to(dev))
        assert float((after - before).abs().max()) <= 2e-3 * float(before.abs().max())
        assert np.abs(after.cpu().numpy() - z[tag + "_logits_smoothed"]).max() <= 2e-3 * np.abs(z[tag + "_logits_smoothed"]).max()


def test_let_temporary_and_inplace_weights_vs_reference(dev):
    """smooth_lm_temporary / smooth_lm_inplace (algorithm.py:147-233) on a sim-quantised decoder layer: every temp_weight /
    temp_bias and the folded weights against the reference's."""
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import DecoderLayer, LlamaShape
    z = load_npz("smooth_cases.npz")
    layer = DecoderLayer(LlamaShape(hidden=64, layers=1, heads=4, kv_heads=4, head_dim=16, ffn=96, vocab=50, eps=1e-5, max_pos=64))
    for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj, layer.mlp.w1, layer.mlp.w2,
                layer.mlp.w3):
        lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features))
    sd = {k.split("|", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith("let|sd|") and "rotary" not in k}
    layer.load_state_dict(sd)
    layer = layer.to(dev)
    mq.create_sim_qmodel(layer, mq.QuantConfig(bitwidth=8, is_per_channel=True), mq.QuantConfig(bitwidth=8))
    for k in z.files:
        if k.startswith("let|param|"):
            layer.register_parameter(k.split("|", 2)[2], torch.nn.Parameter(T(z[k], dev)))
    cfg = type("Cfg", (), dict(shared_attention_norm=False, num_linears_per_mlp=3))()
    mq.smooth_lm_temporary(layer, cfg, True, use_shift=True)
    mods = dict(layer.named_modules())
    n = 0
    for k in z.files:
        if k.startswith("let|temp_weight|") or k.startswith("let|temp_bias|"):
            kind, name = k.split("|")[1], k.split("|", 2)[2]
            got = getattr(mods[name], kind).detach().cpu().numpy()
            assert np.allclose(got, z[k], rtol=2e-6, atol=1e-7), k
            n += 1
    assert n >= 16 and all(m.use_temporary_parameter for m in mods.values() if isinstance(m, mq.QLinear))
    mq.smooth_lm_inplace(layer, cfg, True, use_shift=True)
    for k in z.files:
        if k.startswith("let|inplace|"):
            name = k.split("|", 2)[2]
            got = layer.state_dict()[name].cpu().numpy()
            assert np.allclose(got, z[k], rtol=3e-6, atol=2e-7), name
    assert not any(m.use_temporary_parameter for m in mods.values() if isinstance(m, mq.QLinear))


# ---- f1: integer chain inside the gated FFN ------------------------------------------------------------------------------------
def _ffn(dev, act="silu", hidden=1024, ffn=5632, rows=1536, seed=0):
    """A reference-shaped gated FFN (w1 / w2 / w3 / act_fn) under the W8A8 recipe with static ranges."""
    import mobilequant_amd as mq

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w1 = torch.nn.Linear(hidden, ffn, bias=False)
            self.w2 = torch.nn.Linear(ffn, hidden, bias=False)
            self.w3 = torch.nn.Linear(hidden, ffn, bias=False)
            self.act_fn = torch.nn.SiLU() if act == "silu" else torch.nn.GELU()

        def forward(self, x):
            return self.w2(self.act_fn(self.w1(x)) * self.w3(x))
    torch.manual_seed(seed)
    m = MLP().to(dev)
    x = torch.randn(1, rows, hidden, device=dev)
    with torch.no_grad():
        a, b = m.w1(x), m.w3(x)
        y1 = m.act_fn(a)
        p = y1 * b
        out = m.w2(p)
    rng = lambda t: [float(t.min()), float(t.max())]       # noqa: E731
    act_dict = {"w1": {"input": rng(x), "output": rng(a)}, "w3": {"input": rng(x), "output": rng(b)},
                "act_fn": {"input": rng(a), "output": rng(y1)}, "w2": {"input": rng(p), "output": rng(out)}}
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(m, a8, a8)
    m.w2.weight_quantizer.qcfg.is_per_channel = True          # ptq/mobilequant.py:180-182
    m.w2.output_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(m, act_dict, "buffer")
    mq.wire_integer_inputs(m)
    return m.requires_grad_(False), x, act_dict


@pytest.mark.parametrize("act", ["silu", "gelu"])
def test_fused_gated_mlp_equals_module_chain(dev, act):
    """fuse_gated_mlp: pair GEMM (8-bit output indices of w1 / w3 in one launch) -> gated activation kernel on the indices ->
    w2 GEMM == the chain of Q-modules on their integer paths, bit for bit; and each new kernel against its composite."""
    import mobilequant_amd as mq
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    m, x, act_dict = _ffn(dev, act)
    with torch.no_grad():
        chain = m(x)
        a_fp, b_fp = m.w1(x), m.w3(x)                       # fake-quantised fp32 outputs of the integer path
        assert mq.fuse_gated_mlp(m) == 1 and mq.fuse_gated_mlp(m) == 0
        calls = []
        real_pair, real_gate, real_look = ops.int8_linear_pair, ops.gated_act_quant, ops.gated_lookup
        ops.int8_linear_pair = lambda *a, **k: (calls.append("pair"), real_pair(*a, **k))[1]
        ops.gated_act_quant = lambda *a, **k: (calls.append("gate"), real_gate(*a, **k))[1]
        ops.gated_lookup = lambda *a, **k: (calls.append("lookup"), real_look(*a, **k))[1]
        try:
            fused = m(x)                                    # the gated step through its 256 x 256 table (built on first use)
            m.gated_table = False
            fused_kernel = m(x)                             # ... and through the per-element kernel
            m.gated_table = True
        finally:
            ops.int8_linear_pair, ops.gated_act_quant, ops.gated_lookup = real_pair, real_gate, real_look
        assert calls == ["pair", "lookup", "pair", "gate"]
        assert torch.equal(fused, chain) and torch.equal(fused_kernel, chain)
        # the residual variant (llama.fuse_decoder_layer): resid + mlp(x) with the add inside w2's GEMM store
        r = torch.randn_like(chain)
        seen = []
        real_lin = ops.int8_linear
        ops.int8_linear = lambda *a, **k: (seen.append(k.get("resid") is not None), real_lin(*a, **k))[1]
        try:
            with_resid = m(x, resid=r)
        finally:
            ops.int8_linear = real_lin
        assert seen == [True]
        assert torch.equal(with_resid, r + chain)
        m.fused_mode = "off"
        assert torch.equal(m(x), chain)
        # the pair GEMM's indices are the indices of the two single GEMMs' fake-quantised outputs
        o1, o3 = m.w1.output_quantizer, m.w3.output_quantizer
        grid, a_q, a_rs, a_shift, tiled_rows, _ = m.w1._input_image(x, m.w1.weight)
        assert tiled_rows == x.shape[1]
        halves = []
        for lin in (m.w1, m.w3):
            plan = lin._epilogue_vectors(lin._weight_plan(lin.weight), grid, a_shift, lin.weight.shape[1])
            halves.append(dict(w=plan["w"], alpha=plan["alpha"], w_zp=plan["w_zp"], col_term=plan["col_term"], bias=None,
                               out_scale=lin.output_quantizer.scale, out_offset=lin.output_quantizer.offset))
        ia, ib = ops.int8_linear_pair(a_q, tiled_rows, a_rs, halves[0], halves[1], out_dtype=MQ_U8)
        for idx, fp, oq in ((ia, a_fp, o1), (ib, b_fp, o3)):
            deq = (idx.float() - oq.offset) * oq.scale
            assert torch.equal(deq.view_as(fp), fp)
        # gated kernel on indices == on the fp32 values == composite Q-modules + mq_quantize
        iq2 = m.w2.input_quantizer
        og = (iq2.scale, iq2.offset, iq2.qmin, iq2.qmax)
        mid = None if act == "gelu" else (m.act_fn.input2_quantizer.scale, m.act_fn.input2_quantizer.offset, 0.0, 255.0)
        ao = m.act_fn.output_quantizer
        kw = dict(mid_grid=mid, act_grid=(ao.scale, ao.offset, ao.qmin, ao.qmax), q_shift=128)
        q_i, rs_i = ops.gated_act_quant(ia, ib, act, og, a_grid=(o1.scale, o1.offset), b_grid=(o3.scale, o3.offset), **kw)
        q_f, rs_f, y_f = ops.gated_act_quant(a_fp.reshape(ia.shape), b_fp.reshape(ib.shape), act, og, want_y=True, **kw)
        assert torch.equal(q_i, q_f) and torch.equal(rs_i, rs_f)
        prod = m.act_fn(a_fp) * b_fp
        assert torch.equal(y_f.view_as(prod), prod)
        q_ref, rs_ref, _ = iq2.quantize_to_int(prod.reshape(ia.shape), want_row_sum=True)
        assert torch.equal(q_i, q_ref) and torch.equal(rs_i, rs_ref)


def test_pair_gemm_rejects_what_it_does_not_serve(dev):
    import mobilequant_amd as mq
    from mobilequant_amd import ops
    one = torch.ones(1, device=dev)
    a = torch.zeros(2048, 512, dtype=torch.int8, device=dev)
    w = torch.zeros(5632, 512, dtype=torch.int8, device=dev)
    v = torch.zeros(5632, device=dev)
    zi = torch.zeros(5632, dtype=torch.int32, device=dev)
    half = dict(w=w, alpha=v, w_zp=zi, col_term=zi, bias=None, out_scale=one, out_offset=one)
    with pytest.raises(mq._lib.MobileQuantLibraryError, match="not served"):
        ops.int8_linear_pair(a, 2048, None, half, half)                 # K = 512 < 768


# ---- f2: the decode step against the reference's real model --------------------------------------------------------------------
def _decode_model(dev):
    import json
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    z = load_npz("decode_case.npz")
    m = LlamaForCausalLM(LlamaShape(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=96, eps=1e-5, max_pos=64)).eval()
    sd = {}
    for k in z.files:
        if k.startswith("sd|") and "rotary" not in k:
            name = k[3:]
            sd[name[len("model."):] if name.startswith("model.") else name] = torch.from_numpy(z[k])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("cos" in k or "sin" in k for k in missing)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg"]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    return m.requires_grad_(False), z


def test_decode_engine_reproduces_the_reference_model_token_by_token(dev):
    """DecodeEngine (5 fused launches per layer, static KV cache, position in device memory) fed a 40-token sequence one token at
    a time must reproduce, at every position, the logits the reference's REAL W8A8-simulated HFForCausalLM computes for the whole
    sequence (tests/golden/decode_case.npz), within the budget the prefill kernels state: every quantizer is within one grid step
    on a vanishing fraction of elements, which moves a logit by a small fraction of the logit span; and it must agree with this
    package's own module-graph forward (the prefill path) much more tightly.  Eager steps and the captured-graph replay are
    identical."""
    from mobilequant_amd.decode import DecodeEngine
    m, z = _decode_model(dev)
    ids = torch.from_numpy(z["ids"]).long()
    ref = z["logits_w8a8"][0]
    span = float(np.ptp(z["logits_fp"]))
    with torch.no_grad():
        ours_prefill = m(ids[None].to(dev))[0].cpu().numpy()
    eng = DecodeEngine(m, cache_len=64)
    eager = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids])
    assert int(eng.pos.item()) == len(ids)
    eng.reset()
    eng.capture()
    replay = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids])
    assert np.array_equal(eager, replay)
    d_ref = np.abs(eager - ref)
    d_pre = np.abs(eager - ours_prefill)
    # against the reference (the fixture's own |w8a8 - fp| is ~0.06 of the span).  Bars = twice what is observed (tools/observe_bars.py,
    # round 3: max 0.0097, 99th percentile 0.0029, median 4e-7 of the span, argmax 40 / 40): a systematic one-step bias in any
    # quantizer moves the MEDIAN by orders of magnitude, a single flipped 8-bit index upstream only the tail
    def bars(d, what):
        d = d / span
        assert d.max() <= 0.02 and np.quantile(d, 0.99) <= 0.006 and np.median(d) <= 5e-6, (what, d.max(), np.quantile(d, 0.99), np.median(d))
    bars(d_ref, "decode vs reference")
    assert (eager.argmax(-1) == ref.argmax(-1)).mean() >= 0.975
    # against the prefill kernels of this package: same arithmetic, other summation orders (observed: the same tail, median 3e-8)
    bars(d_pre, "decode vs prefill")
    # greedy generation runs end to end and continues from the context
    out = eng.generate(ids[:8].tolist(), 6)
    assert out[:8] == ids[:8].tolist() and len(out) == 14 and all(0 <= t < 96 for t in out)
    # context encoding in one prefill forward (sim_model.py:176-193), with and without the fused decoder-layer pass: the logits behind
    # the context and those of the following steps stay within the same budget of the reference's
    from mobilequant_amd import llama
    for fused in (False, True):
        if fused:
            assert llama.fuse_decoder_layer(m) == 2
        eng.reset()
        n_ctx = 24
        lg = eng.prefill(ids[:n_ctx].tolist()).cpu().numpy().copy()
        assert int(eng.pos.item()) == n_ctx
        rest = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids[n_ctx:]])
        got = np.concatenate([lg[None], rest])
        want = ref[n_ctx - 1:]
        bars(np.abs(got - want), f"prefill(fused={fused}) + steps vs reference")
        bars(np.abs(got - eager[n_ctx - 1:]), f"prefill(fused={fused}) + steps vs steps only")
    assert eng.generate(ids[:8].tolist(), 6, prefill=True)[:8] == ids[:8].tolist()


def test_decode_gemv_modes_against_prefill_kernels(dev):
    """mq_decode_gemv's fused prologues / epilogues against the prefill kernels on the same row: NORM + segments == QRMSNorm ->
    q/k/v QLinears; GATE == w1 / w3 -> QSiLU -> product -> w2 input quantizer; INT8 + residual == w2 + add."""
    from mobilequant_amd.decode import DecodeEngine
    m, z = _decode_model(dev)
    eng = DecodeEngine(m, cache_len=64, launches=5)          # the five-launch chain: phases[0] / [3] / [4] are its q|k|v, gate and w2 launches
    layer = m.layers[0]
    x = torch.randn(256, device=dev) * 2
    eng.x.copy_(x)
    st = torch.cuda.current_stream().cuda_stream
    import ctypes
    from mobilequant_amd import _lib
    with torch.no_grad():
        # phase 1
        _lib.call("mq_decode_gemv", ctypes.byref(eng.phases[0][1]), st)
        h = layer.input_layernorm(x[None, None])
        want = torch.cat([layer.self_attn.q_proj(h), layer.self_attn.k_proj(h), layer.self_attn.v_proj(h)], dim=-1).flatten()
        lsb = max(float(l.output_quantizer.scale) for l in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj))
        d = (eng.qkv - want).abs()
        assert float(d.max()) <= lsb * 1.001 and float((d == 0).float().mean()) > 0.97
        # phases 4 + 5 (the residual stream is eng.x)
        eng.x.copy_(x)
        _lib.call("mq_decode_gemv", ctypes.byref(eng.phases[3][1]), st)
        g = layer.post_attention_layernorm(x[None, None])
        p = layer.mlp.act_fn(layer.mlp.w1(g)) * layer.mlp.w3(g)
        q_ref, _, _ = layer.mlp.w2.input_quantizer.quantize_to_int(p.reshape(1, -1))
        dq = (eng.gate_q.to(torch.int32) - q_ref.flatten().to(torch.int32)).abs()
        assert int(dq.max()) <= 1 and float((dq == 0).float().mean()) > 0.97
        _lib.call("mq_decode_gemv", ctypes.byref(eng.phases[4][1]), st)
        image = (eng.gate_q.float() + 128 - layer.mlp.w2.input_quantizer.offset) * layer.mlp.w2.input_quantizer.scale
        layer.mlp.w2.int8_mode = "off"
        want2 = x + layer.mlp.w2(image[None, None]).flatten()
        lsb2 = float(layer.mlp.w2.output_quantizer.scale)
        assert float((eng.x - want2).abs().max()) <= lsb2 * 1.001


def test_decode_engine_generic_head_dim_matches_module_graph(dev):
    """head_dim 32 takes the attention kernel's generic (non-vectorised) mapping: the engine must agree with this package's own
    module-graph forward on a random-init model calibrated by this package."""
    import mobilequant_amd as mq
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.decode import DecodeEngine
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from toy_models import apply_mixed_precision
    m = LlamaForCausalLM(LlamaShape(hidden=256, layers=2, heads=8, kv_heads=2, head_dim=32, ffn=512, vocab=64, max_pos=64))
    m.reset_parameters(seed=4, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 64, (1, 24), generator=g)
    act = get_act_range(m, [ids, torch.randint(0, 64, (1, 24), generator=g)])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(m, a8, a8)
    apply_mixed_precision(m, mq)
    mq.set_scale_and_offset(m, act, "buffer")
    with torch.no_grad():
        want = m(ids.to(dev))[0].cpu().numpy()
    eng = DecodeEngine(m, cache_len=64)
    got = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids[0]])
    span = float(np.ptp(want))
    d = np.abs(got - want)
    assert d.max() <= 0.05 * span and np.median(d) <= 0.001 * span and (d <= 0.01 * span).mean() >= 0.97, (d.max() / span, np.median(d) / span)


def test_integer_path_runs_under_inference_mode(dev):
    """Cache keys must not read Tensor._version of inference tensors: norm -> q/k/v chain and a QLinear with its own
    input quantizer give the same results under torch.inference_mode() as under no_grad."""
    import mobilequant_amd as mq
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm
    a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
    torch.manual_seed(0)
    norm = mq.QRMSNorm.from_float(HFRMSNorm(256).to(dev), a16, a16, a8).requires_grad_(False)
    norm.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
    q = mq.QLinear.from_float(torch.nn.Linear(256, 192, bias=False).to(dev), a8, a8, a8).requires_grad_(False)
    q.input_quantizer = None
    q.set_scale_offset({"output": [-3.0, 3.0]}, "buffer")
    w2 = mq.QLinear.from_float(torch.nn.Linear(192, 256, bias=True).to(dev), a8, a8, a16).requires_grad_(False)
    w2.set_scale_offset({"input": [-3.0, 3.0], "output": [-3.0, 3.0]}, "buffer")
    x = torch.randn(2, 40, 256, device=dev)
    with torch.no_grad():
        want = w2(q(norm(x)))
    with torch.inference_mode():
        h = norm(x)
        assert h.is_inference() and q._int8_ready(h, q.weight)
        got = w2(q(h))
        got2 = w2(q(norm(x)))
    assert torch.equal(got, want) and torch.equal(got2, want)


# ---- a10 in context: fused quantized prefill attention ------------------------------------------------------------------------
def _attention_case(S, heads, kv_heads, seed, qk_out_bits=16, pv_out_bits=8):
    rng = np.random.default_rng(seed)
    D = 64
    q = rng.standard_normal((S, heads * D), dtype=np.float32) * 1.5
    k = rng.standard_normal((S, kv_heads * D), dtype=np.float32) * 1.5
    v = rng.standard_normal((S, kv_heads * D), dtype=np.float32)
    inv = 1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float32) / D))
    ang = np.outer(np.arange(S, dtype=np.float32), inv).astype(np.float32)
    ang = np.concatenate((ang, ang), axis=-1)
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)

    def mk(bits, lo, hi):
        o = O.QuantizerOracle(bitwidth=bits)
        o.set_from_minmax(np.float32(lo), np.float32(hi))
        return o
    qr, kr = O.rope_rotate_half(q.reshape(S, heads, D).transpose(1, 0, 2), cos, sin), O.rope_rotate_half(
        k.reshape(S, kv_heads, D).transpose(1, 0, 2), cos, sin)
    sc = np.einsum("hsd,htd->hst", qr, np.repeat(kr, heads // kv_heads, axis=0))
    qk = (mk(8, qr.min(), qr.max()), mk(8, kr.min(), kr.max()), mk(qk_out_bits, sc.min(), sc.max()) if qk_out_bits else None)
    pv = (mk(16, 0.0, 1.0), mk(8, v.min(), v.max()), mk(pv_out_bits, -0.8 * np.abs(v).max(), 0.8 * np.abs(v).max()) if pv_out_bits else None)
    return q, k, v, cos, sin, qk, pv


def _grid_of(o, dev):
    if o is None:
        return None
    return (torch.tensor([float(o.scale)], device=dev), torch.tensor([float(o.offset)], device=dev), float(o.qmin), float(o.qmax))


@pytest.mark.parametrize("S,heads,kv_heads,qk_out_bits,pv_out_bits", [(64, 2, 1, 16, 8), (100, 2, 2, 16, 8), (192, 4, 2, 16, 8), (256, 4, 4, 16, 16), (128, 4, 2, 0, 0)])
def test_attention_quant_vs_oracle(dev, S, heads, kv_heads, qk_out_bits, pv_out_bits):
    """mq_attention_quant against the numpy restatement of hf_model.py:486-534 with its two QMatMuls.  The integer contractions are
    exact where the reference's fp32 matmuls round, and the kernel's quantizers use reciprocal multiplies: a 16-bit score index may
    sit one step off on a vanishing fraction of elements, so the output is compared to one step of its own grid."""
    from mobilequant_amd import ops
    q, k, v, cos, sin, qk, pv = _attention_case(S, heads, kv_heads, seed=S + heads, qk_out_bits=qk_out_bits, pv_out_bits=pv_out_bits)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)
    got = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids).cpu().numpy()
    assert got.shape == want.shape and np.isfinite(got).all()
    diff = np.abs(got - want)
    span = float(want.max() - want.min())
    if pv_out_bits == 8:
        step = float(pv[2].scale)
        assert diff.max() <= 1.001 * step, (diff.max(), step)
        assert (diff > 0.5 * step).mean() < 0.02
    else:
        assert diff.max() <= 2e-3 * span, (diff.max(), span)
    assert np.median(diff) <= 2e-4 * span


def test_fuse_attention_matches_module_chain_and_the_reference_logits(dev):
    """fuse_attention(model) on the 2-layer llama graph of decode_case.npz (the reference's REAL W8A8-simulated HFForCausalLM and
    its logits for a 40-token sequence): prefill with the fused attention against (i) this package's chain of QMatMul modules
    and (ii) the reference's logits, at the budget the decode test states; turning the fused mode off restores the chain."""
    from mobilequant_amd import llama
    m, z = _decode_model(dev)
    ids = torch.from_numpy(z["ids"]).long().to(dev).view(1, -1)
    ref = z["logits_w8a8"][0]
    with torch.no_grad():
        base = m(ids)
        assert llama.fuse_attention(m) == 2 and llama.fuse_attention(m) == 0
        calls = []
        from mobilequant_amd import ops
        real = ops.attention_quant
        ops.attention_quant = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            fused = m(ids)
        finally:
            ops.attention_quant = real
        assert len(calls) == 2
        # q | k | v as one segmented GEMM (uint8 indices, dequantised in the attention prep) == three GEMMs writing fp32: bit for bit
        seg = []
        real_seg = ops.int8_linear_segmented
        ops.int8_linear_segmented = lambda *a, **k: (seg.append(1), real_seg(*a, **k))[1]
        try:
            again = m(ids)
        finally:
            ops.int8_linear_segmented = real_seg
        assert len(seg) == 2 and torch.equal(again, fused)
        for mod in m.modules():
            if isinstance(mod, llama.Attention):
                mod.fuse_qkv = False
        assert torch.equal(m(ids), fused)
        for mod in m.modules():
            if isinstance(mod, llama.Attention):
                mod.fused_mode = "off"
        assert torch.equal(m(ids), base)
    span = float(np.ptp(ref))
    for name, other in (("chain", base[0].cpu().numpy()), ("reference", ref)):
        d = np.abs(fused[0].cpu().numpy() - other)
        assert d.max() <= 0.05 * span and np.median(d) <= 0.001 * span and (d <= 0.01 * span).mean() >= 0.97, (
            name, d.max() / span, np.median(d) / span)


def test_fuse_attention_longer_prefill_with_cache(dev):
    """S = 200 (padded to 256 inside the op), batch 2, static KV cache written on the side: fused == chain within the budget, and the
    cache contents equal the chain's."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama, ops
    from mobilequant_amd.calibration import get_act_range
    from toy_models import apply_mixed_precision
    m = llama.LlamaForCausalLM(llama.LlamaShape(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=64, max_pos=256))
    m.reset_parameters(seed=5, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, 64, (2, 200), generator=g)
    act = get_act_range(m, [ids[:1], ids[1:]])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(m, a8, a8)
    apply_mixed_precision(m, mq)
    mq.set_scale_and_offset(m, act, "buffer")
    mq.wire_integer_inputs(m)
    ids = ids.to(dev)
    with torch.no_grad():
        c0 = m.new_cache(2, 256, device=dev)
        base = m(ids, cache=c0)
        llama.fuse_attention(m)
        c1 = m.new_cache(2, 256, device=dev)
        fused = m(ids, cache=c1)
        assert llama.fuse_decoder_layer(m) == 2          # + residual adds inside the o_proj / w2 GEMM stores: the same fp32 add
        calls = []
        real = ops.int8_linear
        ops.int8_linear = lambda *a, **k: (calls.append(k.get("resid") is not None), real(*a, **k))[1]
        try:
            whole = m(ids, cache=m.new_cache(2, 256, device=dev))
        finally:
            ops.int8_linear = real
        assert sum(calls) == 4                           # o_proj and w2 of both layers
        assert torch.equal(whole, fused)
        m.layers[0].fused_mode = "off"
        assert torch.equal(m(ids, cache=m.new_cache(2, 256, device=dev)), fused)
        m.layers[0].fused_mode = "auto"
    with torch.inference_mode():                         # inference tensors have no version counter: every cache key must cope
        assert torch.equal(m(ids), whole)
    assert torch.equal(c0[0][0], c1[0][0]) and torch.equal(c0[0][1], c1[0][1])      # layer 0: same inputs -> same cache
    span = float(base.max() - base.min())
    d = (fused - base).abs()
    assert float(d.max()) <= 0.05 * span and float(d.median()) <= 0.001 * span, (float(d.max()) / span, float(d.median()) / span)


@pytest.mark.parametrize("S,tiled", [(128, False), (100, False), (128, True), (100, True)])
def test_attention_quant_int8_image_for_o_proj(dev, S, tiled):
    """The optional int8 output image (o_proj's input on pv_bmm's output grid): stored index - 128 must be exactly the index of the
    fp32 output on that grid, in the row-major or the fragment-blocked layout, at a row offset inside a larger image, with row sums;
    rows outside [row0, row0 + S) stay untouched."""
    from mobilequant_amd import ops
    heads, kv_heads = 4, 2
    q, k, v, cos, sin, qk, pv = _attention_case(S, heads, kv_heads, seed=9)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)       # noqa: E731
    K, row0, rows = heads * 64, 32, 32 + S + 7
    img = torch.full(((rows + 15) // 16 * 16, K), 77, dtype=torch.int8, device=dev)
    rs = torch.full((rows,), -5, dtype=torch.int32, device=dev)
    out = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, image=(img, rs, row0, 128, tiled))
    only = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, image=(img.clone(), rs.clone(), row0, 128, tiled),
                               want_out=False)
    assert only is None
    sc, of = np.float32(pv[2].scale), np.float32(pv[2].offset)
    idx = np.rint(out.cpu().numpy() / sc + of).astype(np.int32)
    assert np.array_equal(((idx.astype(np.float32) - of) * sc).astype(np.float32), out.cpu().numpy())       # out sits on the grid
    want = np.full((img.shape[0], K), 77, dtype=np.int8)
    want[row0:row0 + S] = (idx - 128).astype(np.int8)
    got = img.cpu().numpy()
    if tiled:
        # rows of a touched 16-row block that this sequence does not own keep their previous bytes
        assert np.array_equal(got.reshape(-1), tiled_image(want).reshape(-1))
    else:
        assert np.array_equal(got, want)
    rs_want = np.full(rows, -5, dtype=np.int64)
    rs_want[row0:row0 + S] = (idx - 128).sum(1)
    assert np.array_equal(rs.cpu().numpy().astype(np.int64), rs_want)


def test_image_only_norm_outputs_and_their_fallbacks(dev):
    """QRMSNorm.forward_images (used by llama.fuse_decoder_layer): only the int8 image is written; an integer consumer takes it from
    the memo, and every non-integer consumer gets the exact fp32 values back through _materialize -- in both layouts."""
    import mobilequant_amd as mq
    from mobilequant_amd.quantization import qmodule as Q
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm
    torch.manual_seed(0)
    a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
    n = mq.QRMSNorm.from_float(HFRMSNorm(256, eps=1e-5).to(dev), a16, a16, a8).requires_grad_(False)
    n.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
    lin = mq.QLinear.from_float(torch.nn.Linear(256, 64, bias=False).to(dev), a8, a8, a8).requires_grad_(False)
    lin.input_quantizer = None
    lin.set_scale_offset({"output": [-3.0, 3.0]}, "buffer")
    x = torch.randn(2, 40, 256, device=dev) * 1.5
    with torch.no_grad():
        y = n(x)
        want = lin(y)
        for layout in ("rowmajor", "tiled"):
            p = n.forward_images(x, layout)
            assert getattr(p, "_mq_image_only", False) and p.shape == y.shape and p.untyped_storage().nbytes() <= 16
            assert torch.equal(Q._materialize(p), y)
            assert torch.equal(lin(p), want)                       # integer path (memo hit, or rebuilt from the other layout)
            lin.int8_mode = "off"
            assert torch.equal(lin(n.forward_images(x, layout)), lin(y))      # simulated path: materialised first
            lin.int8_mode = "auto"
        small = n.forward_images(x[:1, :4], "rowmajor")            # decode-sized: the GEMV reads fp32 -> ordinary forward
        assert not getattr(small, "_mq_image_only", False) and torch.equal(small, n(x[:1, :4]))


def test_segmented_gemm_equals_the_three_linears(dev):
    """mq_w8a8_linear_segmented (q | k | v in one launch, three 8-bit output grids): every column's uint8 index equals the index the
    linear's own mq_w8a8_linear launch writes, at TinyLlama's q|k|v shape and at a ragged one."""
    import mobilequant_amd as mq
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_I8, MQ_U8
    for M, K, Ns, wbits in ((2048, 2048, (2048, 256, 256), 8), (200, 256, (64, 128, 36), 8), (512, 512, (256, 64, 64), 4)):
        g = torch.Generator(device="cpu").manual_seed(M)
        x = torch.randn(M, K, generator=g).to(dev)
        aq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
        aq.set_scale_offset_from_minmax(float(x.min()), float(x.max()), "buffer", dev)
        a_q, a_rs, a_shift = aq.quantize_to_int(x, MQ_I8, want_row_sum=True)
        parts, singles, grids = [], [], []
        for i, N in enumerate(Ns):
            w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
            bias = (torch.randn(N, generator=g) * 0.1).to(dev)
            wq = mq.Quantizer(mq.QuantConfig(bitwidth=wbits, is_per_channel=(i == 1)))
            wq(w)
            if wbits == 4:      # unsigned nibbles (index - qmin), packed two per byte (QLinear._weight_plan)
                qn, colsum = ops.quantize(w, wq.scale.detach(), wq.offset.detach(), wq.qmin, wq.qmax, q_dtype=MQ_U8, shift=wq.qmin, rows=N,
                                          want_row_sum=True)
                w8, wshift = ops.pack_w4(qn), wq.qmin
            else:
                w8, colsum, wshift = wq.quantize_to_int(w, MQ_I8, want_row_sum=True, rows=N)
            alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, a_shift, wq.scale.detach(), wq.offset.detach(), wshift, colsum, K)
            y = torch.nn.functional.linear(x, w, bias)
            oq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
            oq.set_scale_offset_from_minmax(float(y.min()) * (0.7 + 0.2 * i), float(y.max()) * (0.7 + 0.2 * i), "buffer", dev)
            singles.append(ops.int8_linear(a_q, w8, a_rs, alpha, wzp, ct, bias, out_scale=oq.scale, out_offset=oq.offset, out_qmin=0.0,
                                           out_qmax=255.0, out_dtype=MQ_U8, w4=wbits == 4))
            parts.append((w8, alpha, wzp, ct, bias))
            grids.append((oq.scale, oq.offset))
        cat = [torch.cat([p[j] for p in parts]) for j in range(5)]
        ends = np.cumsum(Ns).tolist()
        got = ops.int8_linear_segmented(a_q, cat[0], a_rs, cat[1], cat[2], cat[3], cat[4], ends, grids, w4=wbits == 4)
        assert got.dtype == torch.uint8 and torch.equal(got, torch.cat(singles, dim=1)), (M, K, Ns, wbits)


def _layer_case_model(dev):
    import json
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from seeded import seeded_parameters_
    z = load_npz("layer_case.npz")
    m = LlamaForCausalLM(LlamaShape(hidden=2048, layers=1, heads=32, kv_heads=4, head_dim=64, ffn=5632, vocab=128, eps=1e-5, max_pos=2048)).eval()
    seeded_parameters_(m, std=0.05)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg"]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    return m.requires_grad_(False), z


def test_baseline_sized_layer_against_the_reference_model(dev):
    """ONE TinyLlama-1.1B decoder layer at S = 2048 -- BASELINE.json's shape: the free-running / pair / segmented / residual GEMMs, the
    fused norms, the gated kernel and the fused attention all run at their production shapes -- against the logits the reference's
    REAL W8A8-simulated HFForCausalLM computes for the same ids, weights (tests/seeded.py) and ranges (tests/golden/layer_case.npz,
    every 8th position).  Quantisation itself moves these logits by 6 % of their span (fixture: logits_fp); the HIP paths must sit
    an order of magnitude closer to the reference than that, and the fully fused layer as close as the chain of modules."""
    from mobilequant_amd import llama
    m, z = _layer_case_model(dev)
    ids = torch.from_numpy(z["ids"]).long().to(dev).view(1, -1)
    ref, fp = z["logits_w8a8"], z["logits_fp"]
    span = float(np.ptp(fp))
    quant_noise = float(np.abs(ref - fp).max()) / span
    assert 0.02 < quant_noise < 0.2
    with torch.no_grad():
        chain = m(ids)[0, ::8].cpu().numpy()
        assert llama.fuse_decoder_layer(m) == 1
        fused = m(ids)[0, ::8].cpu().numpy()
    res = {}
    for name, got in (("chain", chain), ("fused", fused)):
        d = np.abs(got - ref) / span
        res[name] = (float(d.max()), float(np.median(d)), float((d <= 0.005).mean()))
    print("layer_case (max, median, share within 0.005 of span):", res, "quantisation noise", quant_noise)
    for name in res:
        # most positions reproduce the reference to fp32 rounding (median ~1e-6 of the span); a position where one 8-bit index flipped
        # upstream moves by up to ~2 % of the span -- a third of what quantisation itself does to it
        mx, med, share = res[name]
        assert mx <= 0.5 * quant_noise and med <= 1e-4 and share >= 0.95, (name, res[name], quant_noise)
    d = np.abs(fused - chain) / span
    assert d.max() <= 0.5 * quant_noise and np.median(d) <= 1e-4, (float(d.max()), float(np.median(d)))


@pytest.mark.parametrize("wbits,rows,hidden,ffn", [(4, 512, 256, 1024), (8, 300, 256, 512), (4, 2048, 2048, 16384)])
def test_fused_gated_mlp_generalised_shapes_and_w4(dev, wbits, rows, hidden, ffn):
    """fuse_gated_mlp beyond the pair kernel's shapes: packed 4-bit weights (the reference's W4A8 recipes), FFN widths that are no
    multiple of 176 (Gemma: 16384) and small batches run w1 / w3 as two index-writing GEMMs, then the same gated kernel and w2 GEMM.
    Bit-identical to the chain of modules on their integer paths, with and without the residual."""
    import mobilequant_amd as mq
    from mobilequant_amd import ops

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            a8, w = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=wbits, is_per_channel=wbits == 4)
            mk = lambda k, n: mq.QLinear.from_float(torch.nn.Linear(k, n, bias=False).to(dev), a8, w, a8)      # noqa: E731
            self.w1, self.w3, self.w2 = mk(hidden, ffn), mk(hidden, ffn), mk(ffn, hidden)
            self.w1.input_quantizer = self.w3.input_quantizer = None
            self.act_fn = mq.QSiLU(None, a8, a8)

        def forward(self, x):
            return self.w2(self.act_fn(self.w1(x)) * self.w3(x))
    torch.manual_seed(wbits + rows)
    m = MLP().requires_grad_(False)
    x = torch.randn(1, rows, hidden, device=dev)
    xq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
    xq.set_scale_offset_from_minmax(float(x.min()), float(x.max()), "buffer", dev)
    with torch.no_grad():
        x = xq(x)                                           # a tagged activation on an 8-bit grid, as the norm would leave it
        m.w1.set_scale_offset({"output": [-2.0, 2.0]}, "buffer"); m.w3.set_scale_offset({"output": [-2.0, 2.0]}, "buffer")
        m.act_fn.set_scale_offset({"output": [-0.3, 2.0]}, "buffer")
        m.w2.set_scale_offset({"input": [-3.0, 3.0], "output": [-3.0, 3.0]}, "buffer")
        chain = m(x)
        assert mq.fuse_gated_mlp(m) == 1
        calls = []
        real_pair, real_look = ops.int8_linear_pair, ops.gated_lookup
        ops.int8_linear_pair = lambda *a, **k: (calls.append("pair"), real_pair(*a, **k))[1]
        ops.gated_lookup = lambda *a, **k: (calls.append("gate"), real_look(*a, **k))[1]
        try:
            fused = m(x)
            r = torch.randn_like(chain)
            with_resid = m(x, resid=r)
        finally:
            ops.int8_linear_pair, ops.gated_lookup = real_pair, real_look
    assert calls == ["gate", "gate"]                        # no pair kernel here, but the fused chain ran
    assert torch.equal(fused, chain) and torch.equal(with_resid, r + chain)


def test_decode_engine_w4a8_matches_module_graph(dev):
    """The reference's deployment mode: packed 4-bit per-channel weights, 8-bit activations.  mq_decode_gemv unpacks the nibbles in
    registers (all five phases, incl. the interleaved w1|w3 gate phase); the engine must agree with this package's own module-graph
    forward (mq_w4a8_linear GEMMs) on a random-init model calibrated by this package, eagerly and from the captured graph."""
    import mobilequant_amd as mq
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.decode import DecodeEngine
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from toy_models import apply_mixed_precision
    m = LlamaForCausalLM(LlamaShape(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=64, max_pos=64))
    m.reset_parameters(seed=6, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 64, (1, 24), generator=g)
    act = get_act_range(m, [ids, torch.randint(0, 64, (1, 24), generator=g)])
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=4, is_per_channel=True), mq.QuantConfig(bitwidth=8))
    apply_mixed_precision(m, mq)
    mq.set_scale_and_offset(m, act, "buffer")
    with torch.no_grad():
        want = m(ids.to(dev))[0].cpu().numpy()
    eng = DecodeEngine(m, cache_len=64)
    assert all(p[1].w4 == 1 for p in eng.phases if p[0] == "gemv")
    got = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids[0]])
    eng.reset()
    eng.capture()
    replay = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids[0]])
    assert np.array_equal(got, replay)
    span = float(np.ptp(want))
    d = np.abs(got - want)
    assert d.max() <= 0.05 * span and np.median(d) <= 0.001 * span and (d <= 0.01 * span).mean() >= 0.97, (d.max() / span, np.median(d) / span)
    # the fused prefill passes on the same W4A8 model: q|k|v as one segmented GEMM, w1 / w3 as the pair launch
    from mobilequant_amd import llama, ops
    seg = []
    real_seg = ops.int8_linear_segmented
    ops.int8_linear_segmented = lambda *a, **k: (seg.append((k.get("w4"), a[1].dtype, int(a[1].max()) <= 15)), real_seg(*a, **k))[1]
    try:
        with torch.no_grad():
            assert llama.fuse_decoder_layer(m) == 2
            fused = m(ids.to(dev))[0].cpu().numpy()
    finally:
        ops.int8_linear_segmented = real_seg
    # 4-bit weights at prefill: the int8 MFMA kernels on the one-byte-per-nibble image (values 0 .. 15), not the packed image
    assert seg == [(False, torch.int8, True)] * 2
    d = np.abs(fused - want)
    assert d.max() <= 0.05 * span and np.median(d) <= 0.001 * span, (d.max() / span, np.median(d) / span)


@pytest.mark.parametrize("act", ["silu", "gelu"])
def test_gated_table_is_the_gated_kernel_for_every_index_pair(dev, act):
    """mq_gated_table / mq_gated_lookup against mq_gated_act_quant on ALL 65 536 (ia, ib) pairs (and a ragged-width random tensor): the
    table is the kernel's arithmetic evaluated once per pair, so images and row sums are identical."""
    from mobilequant_amd import ops
    import mobilequant_amd as mq

    def grid(lo, hi):
        q = mq.Quantizer(mq.QuantConfig(bitwidth=8))
        q.set_scale_offset_from_minmax(lo, hi, "buffer", dev)
        return (q.scale.detach(), q.offset.detach(), q.qmin, q.qmax)
    ga, gb, gmid, gact, gout = grid(-3.0, 2.5), grid(-2.0, 3.0), grid(0.0, 1.0), grid(-0.3, 2.5), grid(-4.0, 5.0)
    mid = gmid if act == "silu" else None
    table = ops.gated_table(act, gout, ga[:2], gb[:2], mid_grid=mid, act_grid=gact, q_shift=128)
    ia = torch.arange(256, device=dev, dtype=torch.uint8).view(256, 1).expand(256, 256).contiguous()
    ib = torch.arange(256, device=dev, dtype=torch.uint8).view(1, 256).expand(256, 256).contiguous()
    for a_idx, b_idx in ((ia, ib), (torch.randint(0, 256, (37, 1008), device=dev, dtype=torch.uint8),
                                    torch.randint(0, 256, (37, 1008), device=dev, dtype=torch.uint8))):
        want_q, want_rs = ops.gated_act_quant(a_idx, b_idx, act, gout, a_grid=ga[:2], b_grid=gb[:2], mid_grid=mid, act_grid=gact, q_shift=128)
        got_q, got_rs = ops.gated_lookup(a_idx, b_idx, table)
        assert torch.equal(got_q, want_q) and torch.equal(got_rs, want_rs)
    assert torch.equal(table.view(256, 256), ops.gated_act_quant(ia, ib, act, gout, a_grid=ga[:2], b_grid=gb[:2], mid_grid=mid, act_grid=gact,
                                                                 q_shift=128)[0])


@pytest.mark.parametrize("tag,wbits,kv_heads,act", [("w4", 4, 2, "silu"), ("w8pc_mha", 8, 4, "silu"), ("w4_geglu_mqa", 4, 1, "gelu"),
                                                    ("stablelm", 8, 4, "silu"), ("gemma", 4, 1, "gelu")])
def test_other_recipes_against_the_reference_model(dev, tag, wbits, kv_heads, act):
    """(w8pc_mha: the configs[2]-style recipe -- 8-bit per-channel weights everywhere -- with full multi-head attention;
    w4_geglu_mqa: gemma-style GeGLU MLP (QGELU) with multi-query attention, W4A8;
    stablelm / gemma: BASELINE.json configs[2] / [3] on their OWN leaf graphs -- nn.LayerNorm -> QLayerNorm, biased q / k / v and 25 %
    rotary; head_dim 256 with heads * head_dim != hidden, GeGLU and scaled embeddings -- from the reference's HFForCausalLM under the
    matching HFConfig switches (hf_config.py:101-179), W8 per-channel / W4 per-channel symmetric.)
    The reference's W4A8 deployment recipe (4-bit per-channel weights, 8-bit activations) on the 2-layer model, against the logits of
    the reference's REAL HFForCausalLM (tests/golden/decode_case_w4.npz; weights regenerated from tests/seeded.py): the module graph on
    the W4 integer GEMMs, the fused prefill passes (segmented W4 q|k|v, two-GEMM gated MLP) and the W4 decode engine token by token."""
    import json
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd import llama
    from mobilequant_amd.decode import DecodeEngine
    from seeded import seeded_parameters_
    z = load_npz(f"decode_case_{tag}.npz")
    from test_llama_host import FAMILY_SHAPES
    shape_kw = FAMILY_SHAPES.get(tag) or dict(hidden=256, layers=2, heads=4, kv_heads=kv_heads, head_dim=64, ffn=512, vocab=96, eps=1e-5,
                                              max_pos=64, hidden_act=act)
    m = llama.LlamaForCausalLM(llama.LlamaShape(**shape_kw)).eval()
    seeded_parameters_(m, std=0.08)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg"]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    m.requires_grad_(False)
    assert all(mod.weight_quantizer.qcfg.bitwidth == wbits and mod.weight_quantizer.qcfg.is_per_channel
               for mod in m.modules() if isinstance(mod, mq.QLinear))
    ids = torch.from_numpy(z["ids"]).long()
    ref, span = z["logits_w4a8"][0], float(np.ptp(z["logits_fp"]))
    noise = float(np.abs(z["logits_w4a8"] - z["logits_fp"]).max()) / span
    assert noise > (0.1 if wbits == 4 else 0.01)         # 4-bit weights move these logits a lot: the bar below is 1/10 of that
    with torch.no_grad():
        chain = m(ids[None].to(dev))[0].cpu().numpy()
    eng = DecodeEngine(m, cache_len=64)
    steps = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids])
    with torch.no_grad():
        assert llama.fuse_decoder_layer(m) == 2
        fused = m(ids[None].to(dev))[0].cpu().numpy()
    # bars = twice the worst observed over the five recipes (tools/observe_bars.py, round 3: max 0.021, 99th percentile 0.010, median
    # 3.6e-6 of the span, argmax 39 / 40) -- the quantisation itself moves these logits by `noise` = 0.02 .. 0.33 of the span
    for name, got in (("chain", chain), ("decode", steps), ("fused", fused)):
        d = np.abs(got - ref) / span
        assert d.max() <= 0.045 and np.quantile(d, 0.99) <= 0.021 and np.median(d) <= 1e-5, (name, float(d.max()), float(np.quantile(d, 0.99)),
                                                                                                   float(np.median(d)), noise)
        assert (got.argmax(-1) == ref.argmax(-1)).mean() >= 0.95, name
