"""Round-4 GPU parity tests (through the C ABI): model-DEPTH parity against the reference's real 22-layer simulated model, the weight
grid life cycle fixed in round 4, and the kernels that changed shape this round."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    import mobilequant_amd._lib as L
    assert L.device_info()["arch"].startswith("gfx950")
    return torch.device("cuda:0")


def _full_depth_model(dev, tag):
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from seeded import seeded_parameters_
    z = load_npz("full_depth_case.npz")
    S = int(z["ids"].shape[0])
    m = LlamaForCausalLM(LlamaShape(hidden=2048, layers=22, heads=32, kv_heads=4, head_dim=64, ffn=5632, vocab=512, eps=1e-5, max_pos=S)).eval()
    seeded_parameters_(m, std=0.02)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    wcfg = mq.QuantConfig(bitwidth=8) if tag == "w8a8" else mq.QuantConfig(bitwidth=4, is_per_channel=True)
    mq.create_sim_qmodel(m, wcfg, mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg_" + tag]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    return m.requires_grad_(False), z


def _stats(logits, ids):
    lg = logits.double()
    nll = -(torch.log_softmax(lg[:-1], -1).gather(1, ids[1:, None])[:, 0])
    _stats.last_nll = nll.cpu().numpy()
    return float(nll.mean()), float(nll.mean().exp()), lg.argmax(-1).cpu().numpy()


@pytest.mark.parametrize("tag", ["w8a8", "w4a8"])
def test_full_depth_22_layer_model_perplexity_and_argmax_vs_the_reference(dev, tag):
    """BASELINE.json: "quantized perplexity within 0.05 of reference".  The reference's acceptance is a whole-model evaluation
    (eval/harness_eval.py:75-108); offline there are no checkpoints, so depth is pinned on the REAL reference HFForCausalLM at
    TinyLlama-1.1B's geometry (22 layers, hidden 2048, 32 / 4 heads, FFN 5632; weights from tests/seeded.py, 512-word vocabulary, 256
    tokens; tests/golden/full_depth_case.npz): its simulated W8A8 and W4A8 forwards give the next-token NLL / perplexity, the argmax of
    every position and every 8th position's logits.

    What the fixture also shows: on this random-weight model the reference does not reproduce ITSELF to 0.05.  The same simulated forward
    with three BLAS threads instead of one (`*_self3`) moves the perplexity by whole units, agrees on the argmax of ~75-80 % of the
    positions and moves the logits by a median 1 % of their span -- fake-quant behind fp32 matmuls flips ~0.25 % of all 8-bit indices
    when the summation order changes, and 22 layers amplify it.  So does the reference's own op sequence executed on this GPU (every
    QLinear on the simulated path: bit-exact fake-quant kernels around the fp32 library GEMM).  An implementation can therefore be held to
    being one more such sample: the chain of Q-modules on the integer kernels, the fused prefill (9 launches per layer) and the decode
    engine fed token by token must each sit as close to the canonical reference run as those two yardsticks do (logit deviations within
    1.25 x the larger median / max, argmax agreement within 0.08 of the smaller, per-position NLL scatter within 1.25 x), and their
    perplexity inside three standard errors of that scatter (0.05 where the scatter allows it -- here it does not: DESIGN.md 3)."""
    from mobilequant_amd import llama
    from mobilequant_amd.decode import DecodeEngine
    m, z = _full_depth_model(dev, tag)
    ids = torch.from_numpy(z["ids"]).long().to(dev)
    ref_nll = z["nll_" + tag]
    ref_ppl = float(np.exp(ref_nll.mean()))
    fp_ppl = float(np.exp(z["nll_fp"].mean()))
    ref_arg, ref_lg = z["argmax_" + tag], z["logits_" + tag]
    span = float(np.ptp(z["logits_fp"]))
    # the reference against itself
    sd = np.abs(z["logits_" + tag + "_self3"] - ref_lg) / span
    self_ = dict(dppl=float(np.exp(z["nll_" + tag + "_self3"].mean()) - ref_ppl), argmax=float((z["argmax_" + tag + "_self3"] == ref_arg).mean()),
                 logit_max=float(sd.max()), logit_median=float(np.median(sd)), nll_scatter=float(np.std(z["nll_" + tag + "_self3"] - ref_nll)))
    res = {}
    import mobilequant_amd as mq
    with torch.no_grad():
        # a second yardstick: the reference's op sequence on THIS machine -- every QLinear on the simulated path (bit-exact HIP fake-quant
        # around the fp32 library GEMM), i.e. the reference with rocBLAS instead of MKL summing its dot products
        for mod in m.modules():
            if isinstance(mod, mq.QLinear):
                mod.int8_mode = "off"
        lg = m(ids.view(1, -1))[0]
        sim = (_stats(lg, ids), lg[::8].float().cpu().numpy(), _stats.last_nll)
        for mod in m.modules():
            if isinstance(mod, mq.QLinear):
                mod.int8_mode = "auto"
        lg = m(ids.view(1, -1))[0]
        res["module chain"] = (_stats(lg, ids), lg[::8].float().cpu().numpy(), _stats.last_nll)
        assert llama.fuse_decoder_layer(m) == 22
        lg = m(ids.view(1, -1))[0]
        res["fused prefill"] = (_stats(lg, ids), lg[::8].float().cpu().numpy(), _stats.last_nll)
        eng = DecodeEngine(m, cache_len=int(ids.numel()))
        rows = []
        for t in ids.tolist():
            eng.step(t)
            rows.append(eng.logits.clone())
        lg = torch.stack(rows)
        res["decode engine"] = (_stats(lg, ids), lg[::8].float().cpu().numpy(), _stats.last_nll)
    def measure(entry):
        (nll, ppl, arg), sub, nlls = entry
        d = np.abs(sub - ref_lg) / span
        return dict(ppl=round(ppl, 4), dppl=round(ppl - ref_ppl, 4), argmax=round(float((arg == ref_arg).mean()), 4),
                    logit_max=round(float(d.max()), 4), logit_median=round(float(np.median(d)), 5),
                    nll_scatter=round(float(np.std(nlls - ref_nll)), 5))
    report = {name: measure(e) for name, e in res.items()}
    gsim = measure(sim)
    yard = {k: max(self_[k], gsim[k]) for k in ("logit_max", "logit_median", "nll_scatter")}
    yard["argmax"] = min(self_["argmax"], gsim["argmax"])
    ppl_bar = max(0.05, 3.0 * ref_ppl * yard["nll_scatter"] / np.sqrt(ref_nll.size))
    print(f"full depth [{tag}]: reference ppl {ref_ppl:.4f} (fp {fp_ppl:.4f}: quantisation moves it by {ref_ppl - fp_ppl:+.3f}); the reference's own "
          f"second run (3 BLAS threads): {({k: round(v, 5) for k, v in self_.items()})}; the reference's op sequence on this GPU's library GEMM: "
          f"{gsim}; perplexity bar {ppl_bar:.3f};", report)
    for name, r in report.items():
        assert abs(r["dppl"]) <= ppl_bar, (tag, name, r, ppl_bar)
        assert r["argmax"] >= yard["argmax"] - 0.08, (tag, name, r, yard)
        assert r["logit_median"] <= 1.25 * yard["logit_median"] and r["logit_max"] <= 1.25 * yard["logit_max"], (tag, name, r, yard)
        assert r["nll_scatter"] <= 1.25 * yard["nll_scatter"], (tag, name, r, yard)


def test_trained_or_loaded_weight_grid_survives_a_run_time_channel_scale(dev):
    """the first forward derives the weight grid and caches it as nn.Parameters; an optimizer step or
    load_state_dict then changes those SAME tensors in place.  set_input_channel_scale() must re-derive only a grid nobody touched
    since -- a trained or loaded grid is the caller's (qmodule.py:262-277 caches, algorithm.py:239-282 trains `quantizer.scale`)."""
    import mobilequant_amd as mq
    torch.manual_seed(5)
    w = torch.randn(64, 128, device=dev) * 0.05
    s = (torch.rand(128, device=dev) + 0.5)
    x = torch.randn(4, 128, device=dev)

    def make():
        lin = torch.nn.Linear(128, 64, bias=False).to(dev)
        lin.weight.data.copy_(w)
        q = mq.QLinear.from_float(lin, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
        q.set_scale_offset({"input": [-4.0, 4.0], "output": [-2.0, 2.0]}, "buffer")
        return q
    # (1) untouched auto grid: re-derived from W * s
    a = make()
    with torch.no_grad():
        a(x)
    auto = float(a.weight_quantizer.scale)
    a.set_input_channel_scale(s)
    assert not a.weight_quantizer._has_grid()
    with torch.no_grad():
        a(x)
    assert float(a.weight_quantizer.scale) != auto
    # (2) stepped in place (what an optimizer does to `weight_quantizer.scale`): kept
    b = make()
    with torch.no_grad():
        b(x)
        b.weight_quantizer.scale.mul_(1.25)
    stepped = float(b.weight_quantizer.scale)
    b.set_input_channel_scale(s)
    assert b.weight_quantizer._has_grid() and float(b.weight_quantizer.scale) == stepped
    # (3) loaded into the existing tensors: kept
    c, src = make(), make()
    with torch.no_grad():
        c(x)
        src(x)
        src.weight_quantizer.scale.mul_(0.5)
    c.load_state_dict(src.state_dict())
    loaded = float(c.weight_quantizer.scale)
    assert loaded == float(src.weight_quantizer.scale)
    c.set_input_channel_scale(s)
    assert c.weight_quantizer._has_grid() and float(c.weight_quantizer.scale) == loaded


def test_dynamic_activation_quantizers_run_on_the_integer_kernels(dev):
    """`is_dynamic` activation quantizers (qmodule.py:262-277; ptq/mobilequant.py:50,166,205) used to send the whole
    QLinear to the simulated path.  Now: a dynamic INPUT grid is min / max -> scale / offset on the device (mq_minmax_tensor ->
    mq_scale_offset_from_minmax, no host read-back) and the quantize / GEMM kernels take it by pointer; a dynamic OUTPUT grid needs the
    output's own range, so the integer GEMM returns fp32 values and the HIP Quantizer follows (the reference's order, qmodule.py:353-357).
    Against the reference's frozen outputs (tests/golden/qlinear_dynamic_cases.npz): every element within one output LSB, > 99.5 %
    (8-bit) / > 90 % (16-bit) identical -- and int8_coverage() must show the integer path was taken, not the fallback."""
    import mobilequant_amd as mq
    from conftest import load_meta, load_npz
    z = load_npz("qlinear_dynamic_cases.npz")
    T = lambda a: torch.from_numpy(a).to(dev)       # noqa: E731
    for m in load_meta(z):
        k = m["id"]
        lin = torch.nn.Linear(m["K"], m["N"], bias=m["bias"]).to(dev)
        with torch.no_grad():
            lin.weight.copy_(T(z[k + "_w"]))
            if m["bias"]:
                lin.bias.copy_(T(z[k + "_b"]))
        ql = mq.QLinear.from_float(lin, mq.QuantConfig(bitwidth=8, is_dynamic=m["in_dyn"]), mq.QuantConfig(bitwidth=8, is_per_channel=m["wpc"]),
                                   mq.QuantConfig(bitwidth=m["out_bits"], is_dynamic=m["out_dyn"])).requires_grad_(False)
        if "input" in m["act"]:
            ql.input_quantizer.set_scale_offset_from_minmax(*m["act"]["input"], "buffer", dev)
        if "output" in m["act"]:
            ql.output_quantizer.set_scale_offset_from_minmax(*m["act"]["output"], "buffer", dev)
        x = T(z[k + "_x"])
        with torch.no_grad():
            assert ql._int8_reason(x, ql.weight) is None, (m["tag"], ql._int8_reason(x, ql.weight))
            y = ql(x)
            y2 = ql(x * 0.5)                      # a second call re-derives the dynamic grids from ITS tensor
        cov = mq.int8_coverage(ql)
        assert cov["simulated_calls"] == 0 and cov["int8_calls"] == 2, (m["tag"], cov["summary"])
        lsb = float(z[k + "_oscale"])
        d = np.abs(y.cpu().numpy() - z[k + "_y"])
        # a dynamic OUTPUT grid is derived from this implementation's own fp32 outputs: its scale may differ from the reference's in the
        # last bit, so "identical" means the same index on a grid that agrees to fp32 rounding (1e-6 of the value range), not equal bits
        same = d <= (1e-6 * float(np.abs(z[k + "_y"]).max()) if m["out_dyn"] else 0.0)
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)
        assert same.mean() > (0.995 if m["out_bits"] == 8 else 0.90), (m["tag"], same.mean())
        if m["in_dyn"] and m["out_dyn"] and not m["bias"]:
            # both grids scale with the tensor: the half-scale call gives half the values (power-of-two scaling is exact in fp32)
            assert torch.allclose(y2, 0.5 * y, rtol=0, atol=0.51 * 0.5 * lsb), m["tag"]
        # the simulated path of the same module (int8_mode off) agrees within the same bars, and is what coverage reports then
        ql.int8_mode = "off"
        with torch.no_grad():
            ys = ql(x)
        cov = mq.int8_coverage(ql, reset=True)
        assert cov["simulated_calls"] == 1 and "int8_mode off" in cov["summary"], cov["summary"]
        ds = (ys - y).abs()
        tol = 1e-6 * float(y.abs().max()) if m["out_dyn"] else 0.0
        assert float(ds.max()) <= lsb * 1.01 and float((ds <= tol).float().mean()) > (0.99 if m["out_bits"] == 8 else 0.85), m["tag"]


def test_int8_coverage_names_the_modules_that_fell_back_and_why(dev):
    """VERDICT r03 "What's weak" 10: an evaluation can land on the simulated path (HIP fake-quant + fp32 library GEMM, ~10x slower)
    without anyone noticing.  int8_coverage(model) lists, per QLinear, how often each path ran and the reason for every fallback."""
    import mobilequant_amd as mq
    a8 = mq.QuantConfig(bitwidth=8)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            mk = lambda k, n, w: mq.QLinear.from_float(torch.nn.Linear(k, n, bias=False).to(dev), a8, w, a8)      # noqa: E731
            self.ok = mk(128, 64, mq.QuantConfig(bitwidth=8))
            self.group = mk(128, 64, mq.QuantConfig(bitwidth=4, is_per_channel=True, group_size=32))
            self.odd = mk(96, 64, mq.QuantConfig(bitwidth=8))

        def forward(self, x):
            return self.ok(x) + self.group(x), self.odd(x[..., :96].contiguous())
    net = Net().requires_grad_(False)
    for m in (net.ok, net.group, net.odd):
        m.set_scale_offset({"input": [-4.0, 4.0], "output": [-3.0, 3.0]}, "buffer")
    x = torch.randn(2, 16, 128, device=dev)
    with torch.no_grad():
        net(x)
        net(x)
    cov = mq.int8_coverage(net)
    assert cov["int8_calls"] == 2 and cov["simulated_calls"] == 4 and cov["simulated_modules"] == ["group", "odd"], cov
    assert "per-group weight grid" in cov["summary"] and "outside mq_w8a8_linear's limits" in cov["summary"], cov["summary"]
    assert mq.int8_coverage(net, reset=True)["int8_calls"] == 2 and mq.int8_coverage(net)["int8_calls"] == 0


@pytest.mark.parametrize("M,N,K,per_row,with_bias,segs", [
    (2048, 5632, 2048, True, False, None),        # TinyLlama w1 / w3 under the W4A8 recipe: 256 x 176 tiles (frw4)
    (2000, 5632, 768, True, True, None),          # ragged M, the shortest K (KT = 6)
    (2048, 16384, 2048, True, False, None),       # Gemma-2B w1 / w3 (BASELINE.json configs[3]): 256 x 128 tiles (frw4_128)
    (2048, 2560, 2048, True, True, (2048, 2304, 2560)),     # q | k | v as one segmented GEMM, three output grids
    (1024, 2048, 5632, False, False, None),       # long K (44 stages), per-tensor 4-bit grid
])
def test_packed_w4_generated_isa_gemm_vs_exact_oracle_every_output(dev, M, N, K, per_row, with_bias, segs):
    """VERDICT r03 item 3 / BASELINE.json configs[3]: packed 4-bit weights on generated ISA (mq_w4a8_linear_tiled: LDS-DMA of the
    mq_pack_w4 image, one ds_read_b128 per 16 columns and stage, nibbles split in registers under MFMA_I32_16x16x64_I8).  The kernel
    consumes a numpy-built packed image (oracle.pack_w4: the documented layout, NOT the pack kernel) and a numpy-built fragment-blocked
    activation image; EVERY output index is compared with the exact integer oracle, and with what the C++ tile kernel (mq_w4a8_linear)
    writes from the same operands."""
    from oracle import mq_oracle as O
    from test_gpu_round2 import T, exact_u8, tiled_image
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_I8, MQ_U8
    F32 = np.float32
    assert ops.gemm_tiled_w4_supported(M, N, K)
    rng = np.random.default_rng(M + N + K)
    qa = rng.integers(0, 256, size=(M, K))
    qw = rng.integers(0, 16, size=(N, K))                       # unsigned nibbles (index - qmin)
    za = int(rng.integers(0, 256))
    zw = rng.integers(0, 16, size=N) if per_row else np.full(N, int(rng.integers(0, 16)))
    sa = F32(0.02)
    sw = (rng.random(N, dtype=F32) * F32(1e-2) + F32(1e-3)) if per_row else np.full(N, F32(7e-3), F32)
    bias = rng.standard_normal(N, dtype=F32) if with_bias else None
    a8 = (qa - 128).astype(np.int8)
    a_t = T(tiled_image(a8), dev)
    packed = T(O.pack_w4(qw, 0), dev)
    rs = T(a8.sum(1).astype(np.int32), dev)
    colsum = T(qw.sum(1).astype(np.int32), dev)
    wsc = T(sw, dev) if per_row else T(sw[:1], dev)
    wof = T(zw.astype(F32), dev) if per_row else T(zw[:1].astype(F32), dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([sa], F32), dev), T(np.array([za], F32), dev), 128, wsc, wof, 0, colsum, K)
    b = T(bias, dev) if bias is not None else None
    acc, pre = O.qlinear_int_exact(qa, za, sa, qw, zw, sw, bias, blas=True)
    ends = list(segs) if segs else [N]
    grids, want, exact = [], np.empty((M, N), np.uint8), np.empty((M, N))
    lo = 0
    for i, hi in enumerate(ends):
        p = pre[:, lo:hi]
        so = F32((np.percentile(p, 99.5) - np.percentile(p, 0.5)) / 255.0 * (1.0 + 0.1 * i))
        oo = F32(np.rint(-np.percentile(p, 0.5) / so))
        grids.append((torch.tensor([float(so)], device=dev), torch.tensor([float(oo)], device=dev)))
        want[:, lo:hi], exact[:, lo:hi] = exact_u8(qa, za, sa, qw[lo:hi], zw[lo:hi], sw[lo:hi], None if bias is None else bias[lo:hi], so, oo)
        lo = hi
    import mobilequant_amd._lib as L
    lib = L.load()
    try:
        # mode 1 (default): packed pieces expanded once per workgroup into the int8 W ring (frw4x); mode 0: per-wave in-register unpack (frw4)
        for mode in (0, 1):
            if lib.mq_gemm_set_w4_mode(mode) != 0:          # the per-wave unpack kernels (a measured negative): experiment builds only
                continue
            got = ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, b, grids, seg_ends=ends if segs else None).cpu().numpy()
            assert got.shape == (M, N)
            bad = got != want
            assert not bad.any(), (mode, int(bad.sum()), np.argwhere(bad)[:8].tolist(), got[bad][:8].tolist(), want[bad][:8].tolist())
    finally:
        lib.mq_gemm_set_w4_mode(1)
    assert np.abs(got.astype(np.float64) - exact).max() <= 1 and (got == exact).mean() > 0.999
    if not segs:
        got_i8 = ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, b, grids, out_dtype=MQ_I8).cpu().numpy()
        assert np.array_equal(got_i8.astype(np.int16) + 128, want.astype(np.int16))
        # the C++ tile kernel on the same operands (row-major activations)
        ref = ops.int8_linear(T(a8, dev), packed, rs, alpha, wzp, ct, b, out_scale=grids[0][0], out_offset=grids[0][1], out_qmin=0.0,
                              out_qmax=255.0, out_dtype=MQ_U8, w4=True).cpu().numpy()
        assert np.array_equal(ref, got)
    # rows past a ragged M are never written
    canary = torch.full((M + 16, N), 77, dtype=torch.uint8, device=dev)
    ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, b, grids, seg_ends=ends if segs else None, out=canary[:M])
    assert np.array_equal(canary[:M].cpu().numpy(), want) and bool((canary[M:] == 77).all())


def test_packed_only_w4_mode_is_the_image_mode_bit_for_bit_at_full_depth(dev):
    """QLinear.w4_prefill = "packed": every 4-bit module holds ONLY the mq_pack_w4 image (0.5 B / weight, shared by prefill and decode).
    In the fused 22-layer W4A8 model q | k | v (one segmented launch) and w1 / w3 then run mq_w4a8_linear_tiled (generated ISA, pieces
    expanded once per workgroup), o_proj / w2 the packed tile kernel mq_w4a8_linear -- the same integer contraction and the same epilogue
    arithmetic as the int8-image kernels of the default mode, so the logits must be IDENTICAL, and no module may fall back to the
    simulated path."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    res = {}
    try:
        for mode in ("image", "packed"):
            mq.QLinear.w4_prefill = mode
            m, z = _full_depth_model(dev, "w4a8")
            ids = torch.from_numpy(z["ids"]).long().to(dev)
            assert llama.fuse_decoder_layer(m) == 22
            with torch.no_grad():
                res[mode] = m(ids.view(1, -1))[0].clone()
            cov = mq.int8_coverage(m)
            assert cov["simulated_calls"] == 0, cov["summary"]
            if mode == "packed":
                lin = m.layers[0].mlp.w1
                plan = lin._weight_plan(lin.weight)
                assert plan["w4"] and plan["w"].dtype == torch.uint8 and plan["w"].shape == (5632, 1024)
            del m
            torch.cuda.empty_cache()
    finally:
        mq.QLinear.w4_prefill = "image"
    assert torch.equal(res["image"], res["packed"]), float((res["image"] - res["packed"]).abs().max())


def test_decode_engine_notices_quantizers_changed_under_it(dev):
    """The engine snapshots every grid into per-launch constants lines and every weight into an integer image: a grid
    changed in place after the engine was built makes grids_stale() true, and the next reset() / prefill() / capture() re-lowers,
    so that generate() is that of an engine built after the change, token for token and logit for logit."""
    from conftest import load_npz
    from test_gpu_round2 import _decode_model
    from mobilequant_amd.decode import DecodeEngine
    m, _ = _decode_model(dev)
    ctx = load_npz("generate_case.npz")["context"].tolist()
    eng = DecodeEngine(m, cache_len=64).capture()
    before = eng.generate(ctx, 6)
    logits_before = eng.logits.clone()
    assert not eng.grids_stale()
    with torch.no_grad():
        m.layers[0].mlp.w2.input_quantizer.scale.mul_(1.7)           # in place: same tensor object, new version
        q = m.layers[1].self_attn.qk_bmm.input2_quantizer            # replaced: the keys' cache grid
        q.set_scale_offset_from_minmax(-3.0, 2.5)
    assert eng.grids_stale()
    after = eng.generate(ctx, 6)                                      # reset() inside picks the change up and re-records the graph
    assert not eng.grids_stale() and eng.graph is not None
    fresh = DecodeEngine(m, cache_len=64)
    assert fresh.generate(ctx, 6) == after
    assert torch.equal(fresh.logits, eng.logits)
    assert not torch.equal(eng.logits, logits_before) or after != before
    with torch.no_grad():
        m.layers[0].mlp.w1.weight.mul_(1.01)                          # a weight image is a snapshot too
    assert eng.grids_stale()
    eng.refresh_grids()
    assert not eng.grids_stale()


def test_image_cache_rejects_out_of_order_and_ragged_non_final_chunks(dev):
    """ImageCache tracks its filled length: a chunk goes where the previous one ended, only the final chunk may be ragged
    (its pad rows sit in the cache behind it) and a one-token chunk is refused with a message instead of a wrong answer; a ragged
    FINAL chunk (128 + 37) still reproduces the single forward bit for bit."""
    import dataclasses
    from test_gpu_round2 import _decode_model
    from mobilequant_amd import llama
    m, _ = _decode_model(dev)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=256))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    ids = torch.randint(3, m.shape.vocab, (1, 165), generator=torch.Generator().manual_seed(9)).to(dev)
    with torch.no_grad():
        assert llama.fuse_decoder_layer(m) == 2
        whole = m(ids)
        cache = m.new_image_cache(1, 192)
        parts = [m(ids[:, :128], cache=cache, pos=0), m(ids[:, 128:], cache=cache, pos=128)]
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(parts, dim=1), whole)
        assert cache[0].filled == 165
        with pytest.raises(RuntimeError, match="only the final chunk"):
            m(ids[:, :8], cache=cache, pos=165)                                  # behind a ragged chunk
        cache = m.new_image_cache(1, 192)
        m(ids[:, :64], cache=cache, pos=0)
        with pytest.raises(RuntimeError, match="holds 64 positions"):
            m(ids[:, 64:128], cache=cache, pos=128)                              # a hole
        with pytest.raises(RuntimeError, match="one-token chunk"):
            m(ids[:, 64:65], cache=cache, pos=64)


def test_reciprocal_division_is_the_ieee_quotient_over_the_admitted_scale_range(dev, tmp_path):
    """tools/div_check.cpp on this GPU: the one-correction reciprocal form every quantizer kernel uses (mq_common.h div_by_scale) gives
    the IEEE quotient's bits on the quantizer's domain for 48 scales in [1e-5, 1e6] and 54 drawn over +-[2^-60, 2^60] (the range
    scale_in_fast_range admits), every 251st fp32 dividend (the full 2^32 sweep is profiles/r04/div_check.log)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                    os.path.join(root, "tools", "div_check.cpp"), "-o", exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe, "251", "48"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "two corrections 0 mismatches, one correction 0" in r.stdout


@pytest.mark.parametrize("scale", [0.0, 1e-41, 3e-20, 5e19, 1e30, float("inf"), float("nan"), -0.05, -2e-25])
def test_public_quantizer_entry_points_take_the_ieee_divide_for_scales_outside_the_fast_range(dev, scale):
    """mq_fake_quant / mq_quantize with a scale the reciprocal form cannot serve (0, denormal, beyond 2^+-60, inf, NaN): the
    result is the reference expression's, evaluated by torch on the CPU in fp32 -- clamp(round_ste(x / s) + o) and its dequantised
    value, NaN where the reference makes NaN -- per tensor and per row, vector and scalar kernels."""
    from mobilequant_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 80, generator=g) * 3
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e30, -1e30, 65504.0, 1.0])
    for shape, sc in (((6, 80), torch.tensor([scale])), ((6, 80), torch.tensor([scale, 0.02, scale, 1.0, 0.5, scale])),
                      ((6, 77), torch.tensor([scale]))):
        xs = x[:, :shape[1]].contiguous()
        off = torch.full_like(sc, 7.0)
        s2, o2 = (sc.view(-1, 1), off.view(-1, 1)) if sc.numel() > 1 else (sc, off)
        t = xs / s2
        idx = torch.clamp((torch.round(t) - t) + t + o2, 0.0, 255.0)                     # qmodule.py:286-287 (clamp keeps NaN)
        want = (idx - o2) * s2
        got = ops.fake_quant(xs.to(dev), sc.to(dev), off.to(dev), 0.0, 255.0).cpu()
        assert torch.equal(torch.nan_to_num(got, nan=12345.0), torch.nan_to_num(want, nan=12345.0)), (scale, shape, sc.numel())
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        q = ops.quantize(xs.to(dev), sc.to(dev), off.to(dev), 0.0, 255.0, q_dtype=ops.MQ_U8)
        q = (q[0] if isinstance(q, tuple) else q).cpu()
        # integer storage has no NaN: the index saturates (rint(t) + o clamped; an overflowed quotient -> qmin / qmax, NaN -> qmin)
        want_i = torch.nan_to_num(torch.clamp(torch.round(t) + o2, 0.0, 255.0), nan=0.0).to(torch.uint8)
        assert torch.equal(q, want_i), (scale, shape, sc.numel())


@pytest.mark.parametrize("rows,cols,bits,sym", [(64, 2048, 4, False), (33, 5632, 4, False), (16, 1024, 8, False), (48, 2048, 4, True),
                                                 (8, 16384, 4, False), (20, 3072, 8, True), (12, 64, 4, False)])
def test_fused_lwc_pass_is_the_reference_expression_forward_bit_for_bit_and_its_autograd(dev, rows, cols, bits, sym):
    """mq_lwc_fake_quant / _backward (one HIP pass per direction) against the reference's expression evaluated by torch on the CPU in
    fp32 with autograd -- amin / amax per row (qmodule.py:263-268), sigmoid(bound) * range (:271-273), the grid (:40-61), round_ste
    and the fake-quant (:17-21, :286-290): forward values and the grid bit-identical; gradients of the weight (and so of a LET
    scale upstream), and of both sigmoid(bound factor) vectors equal up to the association of the row sums.  Rows with tied extremes,
    an all-zero row (scale at CLIPMIN: the clamp blocks the gradient) and a constant row included.  (The module chain the pass
    replaces evaluated alpha / q_max on the GPU as torch does for a Python scalar divisor -- alpha * (1 / q_max) -- and so sat one ulp
    off the reference's CPU arithmetic on some rows; the pass uses the true quotient, like mq_scale_offset_from_minmax.)"""
    from mobilequant_amd.quantization import qmodule as Q
    g = torch.Generator().manual_seed(rows * 7 + cols + bits)
    w0 = (torch.randn(rows, cols, generator=g) * 0.05)
    w0[1, 5] = w0[1, 9] = w0[1].max() + 0.01                       # tie at the maximum
    w0[2, 3] = w0[2, 40] = w0[2, 41] = w0[2].min() - 0.02          # three-way tie at the minimum
    w0[3] = 0.0                                                    # degenerate range
    w0[4] = 0.037                                                  # constant row: every element is both extreme
    col0 = 1.0 + 0.2 * torch.randn(cols, generator=g)              # a LET scale upstream of the quantizer
    gy = torch.randn(rows, cols, generator=g)
    sig_lo0 = torch.sigmoid(torch.linspace(4.5, 0.5, rows)).view(-1, 1)
    sig_hi0 = torch.sigmoid(torch.linspace(1.0, 5.0, rows)).view(-1, 1)
    qmin, qmax = Q._grid_limits(bits, sym)

    def leaves(device):
        return [t.clone().to(device).requires_grad_(True) for t in (col0, sig_lo0, sig_hi0)]
    # the reference's expression, CPU fp32
    col, slo, shi = leaves("cpu")
    temp = w0 * col.view(1, -1)
    lo, hi = slo * temp.amin(-1, keepdim=True), shi * temp.amax(-1, keepdim=True)
    alpha, beta = (torch.maximum(lo.abs(), hi.abs()), 0 * lo) if sym else (hi - lo, lo)
    scale = (alpha / qmax).clamp(min=1e-5, max=1e6)
    offset = -(beta / scale).round()
    t = temp / scale
    want = ((((t.round() - t).detach() + t) + offset).clamp(qmin, qmax) - offset) * scale
    (want * gy).sum().backward()
    # the HIP pass
    colg, slog, shig = leaves(dev)
    y, sc, of = Q._LwcFakeQuantFn.apply(w0.to(dev) * colg.view(1, -1), slog, shig, bits, sym)
    (y * gy.to(dev)).sum().backward()
    assert torch.equal(sc.cpu().view(-1), scale.detach().view(-1)) and torch.equal(of.cpu().view(-1), offset.detach().view(-1) + 0.0)
    assert torch.equal(y.detach().cpu(), want.detach())
    for a, b, name in ((colg.grad, col.grad, "LET scale"), (slog.grad, slo.grad, "sigmoid(lowbound)"), (shig.grad, shi.grad, "sigmoid(upbound)")):
        tol = 3e-5 * float(b.abs().max()) + 1e-7
        assert float((a.cpu() - b).abs().max()) <= tol, (name, float((a.cpu() - b).abs().max()), tol)
    assert float(shig.grad.abs().max()) > 0 and float(colg.grad.abs().max()) > 0


def test_quantizer_in_lwc_mode_takes_the_fused_pass_and_agrees_with_the_module_chain(dev):
    """Quantizer.forward in LWC mode routes per-channel 2-D weights through the fused pass (lwc_fused, default on): same grid
    attributes left behind as the module chain (plain [rows, 1] tensors, a new generation), values within one grid step of the
    chain's (the one-ulp scale difference described above), gradients of the bound factors within 1e-3 relative."""
    import mobilequant_amd as mq
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(96, 2048, generator=g) * 0.05).to(dev)
    gy = torch.randn(96, 2048, generator=g).to(dev)
    res = []
    for fused in (True, False):
        q = mq.Quantizer(mq.QuantConfig(bitwidth=4, is_per_channel=True))
        q.enable_lwc(w)
        q.lwc_fused = fused
        y = q(w)
        (y * gy).sum().backward()
        assert q.scale.shape == (96, 1) and q.offset.shape == (96, 1) and not isinstance(q.scale, torch.nn.Parameter)
        res.append((y.detach(), q.scale.detach(), q.upbound_factor.grad.clone(), q.lowbound_factor.grad.clone()))
    (yf, sf, guf, glf), (ym, sm, gum, glm) = res
    assert float(((sf - sm).abs() / sm).max()) <= 2.5e-7
    assert float((yf - ym).abs().max()) <= float(sm.max()) * 1.001
    assert float(((yf - ym).abs() > 1e-6 * float(ym.abs().max())).float().mean()) < 1e-3      # index flips; the rest moves by an ulp of the scale
    for a, b in ((guf, gum), (glf, glm)):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max())


@pytest.mark.parametrize("heads,S,T,masked", [(2, 64, 64, True), (3, 40, 100, False), (1, 96, 2048, True), (1, 24, 4096, False), (2, 16, 260, True)])
def test_training_attention_probabilities_pass_vs_the_reference_expression_and_its_autograd(dev, heads, S, T, masked):
    """mq_attention_probs_train / _backward -- Q_pv_in(softmax(Q_qk_out(raw) / sqrt(d) + mask)) with learnable 16-bit grids
    (hf_model.py:511-520, qmodule.py:286-290) -- against the same expression evaluated by torch on the CPU with autograd: values
    within one step of the probability grid on < 0.1 % of the elements and identical elsewhere up to an ulp of the softmax (row sums
    associate differently), the gradient of the raw scores within 1e-4 of its maximum, the four grid gradients within 2 %."""
    from mobilequant_amd.quantization import qmodule as Q
    g = torch.Generator().manual_seed(heads * 100 + S + T)
    raw0 = torch.randn(1, heads, S, T, generator=g) * 6.0
    gy = torch.randn(1, heads, S, T, generator=g)
    mask = None
    if masked:
        mask = torch.zeros(S, T)
        mask[torch.arange(S).view(-1, 1) + (T - S) < torch.arange(T).view(1, -1)] = float("-inf")       # causal, last S of T positions
    lim1, lim2 = (0.0, 65535.0), (0.0, 65535.0)
    grids0 = [torch.tensor(v) for v in (48.0 / 65535, 32768.0, 1.0 / 65535 * 0.9, 0.0)]                  # p above 0.9 clamps: both masks exercised
    sqrt_d = 8.0

    def fq(x, s, o, lim):
        t = x / s
        return ((((t.round() - t).detach() + t) + o).clamp(*lim) - o) * s

    def run(device, fused):
        raw = raw0.clone().to(device).requires_grad_(True)
        gr = [t.clone().to(device).requires_grad_(True) for t in grids0]
        m = None if mask is None else mask.to(device)
        if fused:
            p = Q._AttnProbsFn.apply(raw, gr[0], gr[1], gr[2], gr[3], m, lim1, lim2, sqrt_d)
        else:
            a = fq(raw, gr[0], gr[1], lim1) / sqrt_d
            p = fq(torch.softmax(a if m is None else a + m, dim=-1, dtype=torch.float32), gr[2], gr[3], lim2)
        (p * gy.to(device)).sum().backward()
        return p.detach().cpu(), raw.grad.cpu(), [t.grad.cpu() for t in gr]
    p_ref, graw_ref, gg_ref = run("cpu", False)
    p, graw, gg = run(dev, True)
    step = float(grids0[2])
    d = (p - p_ref).abs()
    assert float(d.max()) <= step * 1.001 and float((d > step * 1e-3).float().mean()) < 1e-3
    assert float((graw - graw_ref).abs().max()) <= 1e-4 * float(graw_ref.abs().max())
    for a, b, name in zip(gg, gg_ref, ("d s1", "d o1", "d s2", "d o2")):
        assert abs(float(a) - float(b)) <= 2e-2 * abs(float(b)) + 1e-6 * float(graw_ref.abs().max()) * raw0.numel() ** 0.5, (name, float(a), float(b))
    assert float(gg_ref[2].abs()) > 0 and float(gg_ref[0].abs()) > 0 and float(graw_ref.abs().max()) > 0


def test_training_mode_attention_block_uses_the_fused_probabilities_pass(dev):
    """Attention.forward with gradients wanted runs the score-sized chain as _AttnProbsFn (one saved score tensor instead of four):
    output and every gradient (input, learnable grids of both QMatMuls) agree with the module chain (train_fused = False) within
    the noise of a different softmax summation order; without gradients the module chain runs as before (bit-identical output)."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import get_act_range
    shape = llama.LlamaShape(vocab=64, hidden=256, ffn=512, layers=1, heads=4, kv_heads=2, head_dim=64, max_pos=128)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=5, std=0.08)
    model = model.to(dev).eval()
    ids = torch.randint(0, 64, (1, 96), generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        act = get_act_range(model, [ids])
    mq.create_sim_qmodel(model, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    attn = model.layers[0].self_attn
    attn.qk_bmm.output_quantizer.qcfg.bitwidth = 16
    attn.pv_bmm.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "parameter")
    x0 = model.embed_tokens(ids).detach()
    cos, sin = model.cos[:96], model.sin[:96]
    mask = torch.full((96, 96), float("-inf"), device=dev).triu(1)
    grids = [attn.qk_bmm.output_quantizer.scale, attn.qk_bmm.output_quantizer.offset, attn.pv_bmm.input_quantizer.scale, attn.pv_bmm.input_quantizer.offset]
    gy = torch.randn(1, 96, 256, generator=torch.Generator().manual_seed(4)).to(dev)
    res = []
    for fused in (True, False):
        attn.qk_bmm.train_fused = fused
        x = x0.clone().requires_grad_(True)
        for p in grids:
            p.grad = None
        y = attn(x, cos, sin, mask)
        (y * gy).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in grids]))
    attn.qk_bmm.train_fused = True
    (yf, gxf, ggf), (ym, gxm, ggm) = res
    assert float((yf - ym).abs().max()) <= 2e-2 * float(ym.abs().max())
    assert float((gxf - gxm).abs().max()) <= 2e-2 * float(gxm.abs().max())
    for a, b in zip(ggf, ggm):
        assert abs(float(a) - float(b)) <= 5e-2 * abs(float(b)) + 1e-4
    with torch.no_grad():
        a = attn(x0, cos, sin, mask)
        attn.qk_bmm.train_fused = False
        b = attn(x0, cos, sin, mask)
        attn.qk_bmm.train_fused = True
    assert torch.equal(a, b)


@pytest.mark.parametrize("M,K", [(2048, 2048), (1900, 768), (1552, 5632)])
def test_pair_gemm_persistent_over_the_pair_is_the_two_workgroup_launch_bit_for_bit(dev, M, K):
    """mq_gemm_set_pair_mode(1): one workgroup per tile runs problem 0 then problem 1 (half the workgroups) -- the indices of both
    outputs equal those of the default launch (one workgroup per tile and problem) and of the two single GEMMs, ragged M included."""
    from mobilequant_amd import ops, _lib
    from mobilequant_amd._lib import MQ_U8
    N = 5632
    g = torch.Generator().manual_seed(M + K)
    Mp = (M + 15) // 16 * 16
    a_rm = torch.randint(-128, 128, (Mp, K), generator=g, dtype=torch.int8)
    a_rm[M:] = 0
    a_t = a_rm.view(Mp // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(Mp, K).to(dev)
    rs = a_rm[:M].to(torch.int32).sum(1, dtype=torch.int32).to(dev)
    halves = []
    for i in range(2):
        halves.append(dict(w=torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(dev),
                           alpha=(torch.rand(N, generator=g) * 2e-4 + 1e-5).to(dev), w_zp=torch.randint(-3, 4, (N,), generator=g, dtype=torch.int32).to(dev),
                           col_term=torch.randint(-50000, 50000, (N,), generator=g, dtype=torch.int32).to(dev),
                           bias=(torch.randn(N, generator=g) * 0.1).to(dev) if i else None,
                           out_scale=torch.tensor([0.011 * (i + 1)], device=dev), out_offset=torch.tensor([100.0 + 20 * i], device=dev)))
    lib = _lib.load()
    res = []
    for mode in (0, 1):
        lib.mq_gemm_set_pair_mode(mode)
        try:
            res.append([t.clone() for t in ops.int8_linear_pair(a_t, M, rs, halves[0], halves[1], out_dtype=MQ_U8)])
        finally:
            lib.mq_gemm_set_pair_mode(0)
    torch.cuda.synchronize()
    for a, b, h in zip(res[0], res[1], halves):
        assert torch.equal(a, b)
        single = ops.int8_linear(a_t, h["w"], rs, h["alpha"], h["w_zp"], h["col_term"], h["bias"], out_scale=h["out_scale"], out_offset=h["out_offset"],
                                 out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8, a_tiled_rows=M)
        assert torch.equal(single, a)


@pytest.mark.parametrize("M,N,K,gs,wbits,sym", [(256, 256, 1024, 128, 4, False), (300, 384, 768, 64, 8, False), (2048, 2048, 2048, 128, 4, True),
                                                 (77, 128, 512, 256, 8, True), (130, 5632, 2048, 128, 4, False)])
def test_per_group_weight_grids_on_the_integer_path_exact_and_vs_the_simulated_module(dev, M, N, K, gs, wbits, sym):
    """mq_w8a8_linear_grouped: QLinear with group_size != -1 (qmodule.py:259-260, :292-293).  (i) The kernel against the reference's
    expression evaluated from the SAME integer indices in float64 -- sum_g s_a s_w[n, g] sum_{k in g} (ia - z_a)(iw - o_w[n, g]) + bias --
    to fp32 accumulation accuracy (the integer brackets are exact).  (ii) The module on the integer path against the module on the
    simulated path (fake-quant + fp32 library GEMM): within one step of the 8-bit output grid on < 1 % of the outputs, identical
    elsewhere; int8_coverage() counts the call as integer."""
    import mobilequant_amd as mq
    g = torch.Generator().manual_seed(M + N + K + gs)
    lin = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) * 0.05 * (1 + torch.rand(N, 1, generator=g)))
        lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
    q = mq.QLinear.from_float(lin.to(dev), mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=wbits, is_per_channel=True, group_size=gs, is_symmetric=sym),
                              mq.QuantConfig(bitwidth=8)).requires_grad_(False)
    x = (torch.randn(1, M, K, generator=g) * 1.3).to(dev)
    y_fp = torch.nn.functional.linear(x, q.weight, q.bias)
    q.set_scale_offset({"input": [float(x.min()), float(x.max())], "output": [float(y_fp.min()), float(y_fp.max())]}, "buffer")
    assert q._int8_reason(x, q.weight) == "per-group weight grid" and q._grouped_reason(x, q.weight) is None
    mq.int8_coverage(q, reset=True) if hasattr(mq, "int8_coverage") else None
    y_int = q(x)
    q.int8_mode = "off"
    y_sim = q(x)
    q.int8_mode = "auto"
    oq = q.output_quantizer
    step = float(oq.scale)
    d = (y_int - y_sim).abs()
    assert float(d.max()) <= step * 1.001 and float((d > step * 1e-3).float().mean()) < 1e-2, (float(d.max()) / step, float((d > step * 1e-3).float().mean()))
    # (i) exact: rebuild the indices the kernel used and evaluate the reference expression in float64
    plan, grid = q._grouped_plan(q.weight), q._activation_grid(x)
    a_q, _, a_shift = grid.quantize_to_int(x.reshape(-1, K), mq.quantization.qmodule.MQ_I8)
    ia = a_q.double() + a_shift
    wq = q.weight_quantizer
    G = K // gs
    shift_w = 128 if wq.qmax > 127 else 0
    iw = plan["w"].double() + shift_w
    sw, ow = wq.scale.double().reshape(N, G), wq.offset.double().reshape(N, G)
    a_deq = (ia - grid.offset.double()) * grid.scale.double()
    w_deq = ((iw.reshape(N, G, gs) - ow[:, :, None]) * sw[:, :, None]).reshape(N, K)
    want = a_deq @ w_deq.t() + q.bias.double()
    q.output_quantizer.enable = False
    got = q(x).reshape(M, N).double()
    q.output_quantizer.enable = True
    err = (got - want).abs().max()
    assert float(err) <= 2e-5 * float(want.abs().max()) + 1e-6, float(err)


def test_per_group_weight_grids_vs_the_reference_outputs(dev):
    """tests/golden/qlinear_grouped_cases.npz: the reference's QLinear with group_size 64 / 128 / 256, 4- / 8-bit, symmetric / asymmetric
    weight grids.  The module takes the integer path (mq_w8a8_linear_grouped), derives the reference's per-group scale / offset bit for
    bit, and lands within one output LSB of the reference's outputs on every element, > 99 % identical."""
    import mobilequant_amd as mq
    from conftest import load_meta, load_npz
    z = load_npz("qlinear_grouped_cases.npz")
    T = lambda a: torch.from_numpy(a).to(dev)       # noqa: E731
    for m in load_meta(z):
        k = m["id"]
        lin = torch.nn.Linear(m["K"], m["N"], bias=m["bias"]).to(dev)
        with torch.no_grad():
            lin.weight.copy_(T(z[k + "_w"]))
            if m["bias"]:
                lin.bias.copy_(T(z[k + "_b"]))
        ql = mq.QLinear.from_float(lin, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=m["wbits"], is_per_channel=True, group_size=m["gs"],
                                                                                   is_symmetric=m["sym"]), mq.QuantConfig(bitwidth=8)).requires_grad_(False)
        ql.set_scale_offset(m["act"], "buffer")
        x = T(z[k + "_x"])
        with torch.no_grad():
            y = ql(x)
        cov = mq.int8_coverage(ql, reset=True)
        assert cov["simulated_calls"] == 0 and cov["int8_calls"] == 1, (m["tag"], cov["summary"])
        wq = ql.weight_quantizer
        assert np.array_equal(wq.scale.detach().cpu().numpy().reshape(-1), z[k + "_wscale"].reshape(-1)), m["tag"]
        assert np.array_equal(wq.offset.detach().cpu().numpy().reshape(-1) + 0.0, z[k + "_woffset"].reshape(-1) + 0.0), m["tag"]
        lsb = float(z[k + "_oscale"])
        d = np.abs(y.cpu().numpy() - z[k + "_y"])
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)
        assert (d == 0).mean() > 0.99, (m["tag"], (d == 0).mean())


@pytest.mark.parametrize("M,N,K,per_row,with_bias", [(2048, 2048, 2048, True, False), (2048, 2048, 5632, True, True), (300, 256, 768, False, False),
                                                      (129, 384, 1024, True, True), (1024, 2048, 16384, True, False)])
def test_packed_w4_residual_gemm_is_the_int8_image_residual_gemm_bit_for_bit(dev, M, N, K, per_row, with_bias):
    """mq_w4a8_linear_tiled_residual (generated ISA frw4x_128r: the packed pieces of o_proj / w2 expanded once per workgroup into the int8
    ring, 128 x 128 tiles, x + Q16(linear) in the store) fed a numpy-built packed image (oracle.pack_w4) against
    mq_w8a8_linear_tiled_residual on the one-byte-per-nibble image of the same numbers: the same fp32 bits, ragged M included."""
    from oracle import mq_oracle as O
    from test_gpu_round2 import T, tiled_image
    from mobilequant_amd import ops
    F32 = np.float32
    rng = np.random.default_rng(M + N + K + 1)
    qa = rng.integers(0, 256, size=(M, K))
    qw = rng.integers(0, 16, size=(N, K))
    za = int(rng.integers(0, 256))
    zw = rng.integers(0, 16, size=N) if per_row else np.full(N, int(rng.integers(0, 16)))
    sw = (rng.random(N, dtype=F32) * F32(1e-2) + F32(1e-3)) if per_row else np.full(N, F32(7e-3), F32)
    a8 = (qa - 128).astype(np.int8)
    a_t = T(tiled_image(a8), dev)
    rs = T(a8.sum(1).astype(np.int32), dev)
    colsum = T(qw.sum(1).astype(np.int32), dev)
    wsc = T(sw, dev) if per_row else T(sw[:1], dev)
    wof = T(zw.astype(F32), dev) if per_row else T(zw[:1].astype(F32), dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([0.02], F32), dev), T(np.array([za], F32), dev), 128, wsc, wof, 0, colsum, K)
    b = T(rng.standard_normal(N, dtype=F32), dev) if with_bias else None
    resid = torch.randn(M, N, device=dev)
    pre = (qa - za).astype(np.float64) @ ((qw - zw[:, None]) * sw[:, None]).astype(np.float64).T * 0.02
    span = float(np.percentile(pre, 99.9) - np.percentile(pre, 0.1))
    step = span / 50000.0
    so, oo = torch.tensor([step], device=dev), torch.tensor([float(np.rint(32768.0 - np.median(pre) / step))], device=dev)
    kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0, resid=resid, a_tiled_rows=M)
    want = ops.int8_linear(a_t, T(qw.astype(np.int8), dev), rs, alpha, wzp, ct, b, **kw)
    got = ops.int8_linear(a_t, T(O.pack_w4(qw, 0), dev), rs, alpha, wzp, ct, b, w4=True, **kw)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got, want), float((got - want).abs().max())
    q = torch.round((want - resid) / so + oo)
    assert float(q.min()) < 20000 and float(q.max()) > 45000            # the 16-bit grid is exercised, not saturated


@pytest.mark.parametrize("M,N,K", [(2048, 5632, 2048), (1900, 5632, 768), (2048, 16384, 2048)])
def test_packed_w4_gated_pair_is_the_int8_image_gated_pair_bit_for_bit(dev, M, N, K):
    """mq_w4a8_linear_tiled_gated (w1 on frw4x, w3 on frgw4x: the gate's table lookup in the epilogue of the packed kernel) against
    mq_w8a8_linear_tiled_gated on the one-byte-per-nibble images of the same numbers: w2's input image and its row sums, same bytes."""
    from oracle import mq_oracle as O
    from test_gpu_round2 import T, tiled_image
    from mobilequant_amd import ops
    F32 = np.float32
    rng = np.random.default_rng(M + N + K + 5)
    qa = rng.integers(0, 256, size=(M, K))
    a8 = (qa - 128).astype(np.int8)
    a_t, rs = T(tiled_image(a8), dev), T(a8.sum(1).astype(np.int32), dev)
    za = int(rng.integers(100, 156))
    halves_i8, halves_w4 = [], []
    for i in range(2):
        qw = rng.integers(0, 16, size=(N, K))
        zw = rng.integers(0, 16, size=N)
        sw = rng.random(N, dtype=F32) * F32(1e-2) + F32(1e-3)
        alpha, wzp, ct = ops.linear_epilogue_prepare(T(np.array([0.02], F32), dev), T(np.array([za], F32), dev), 128, T(sw, dev), T(zw.astype(F32), dev), 0,
                                                     T(qw.sum(1).astype(np.int32), dev), K)
        common = dict(alpha=alpha, w_zp=wzp, col_term=ct, bias=T(rng.standard_normal(N, dtype=F32), dev) if i else None,
                      out_scale=torch.tensor([0.9 + 0.3 * i], device=dev), out_offset=torch.tensor([120.0 + 10 * i], device=dev))
        halves_i8.append(dict(w=T(qw.astype(np.int8), dev), **common))
        halves_w4.append(dict(w=T(O.pack_w4(qw, 0), dev), **common))
    table = torch.from_numpy(rng.integers(-128, 128, size=65536).astype(np.int8)).to(dev)
    want_q, want_rs = ops.int8_linear_gated(a_t, M, rs, halves_i8[0], halves_i8[1], table)
    got_q, got_rs = ops.int8_linear_gated(a_t, M, rs, halves_w4[0], halves_w4[1], table, w4=True)
    torch.cuda.synchronize()
    Mp = (M + 15) // 16 * 16
    back = lambda t: t.view(Mp // 16, N // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, N)[:M]      # noqa: E731  (rows past M are padding)
    assert torch.equal(back(got_q), back(want_q)) and torch.equal(got_rs, want_rs)
    assert int(want_q.to(torch.int32).abs().max()) > 100
