"""Round-5 GPU parity tests (through the C ABI): the 0.05-perplexity bar at model depth with NO self-calibrated yardstick, the epilogue
quantizer's flip rate against the reference's divide form, the standalone integer QMatMul, and the advisor's round-4 findings."""
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


def _stable_model(dev, tag):
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from seeded import seeded_contractive_parameters_
    z = load_npz("full_depth_stable_case.npz")
    S = int(z["ids"].shape[0])
    m = LlamaForCausalLM(LlamaShape(hidden=2048, layers=22, heads=32, kv_heads=4, head_dim=64, ffn=5632, vocab=512, eps=1e-5, max_pos=S)).eval()
    seeded_contractive_parameters_(m)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    wcfg = mq.QuantConfig(bitwidth=8) if tag == "w8a8" else mq.QuantConfig(bitwidth=4, is_per_channel=True)
    mq.create_sim_qmodel(m, wcfg, mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg_" + tag]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    return m.requires_grad_(False), z


def _nll(logits, ids):
    lg = logits.double()
    return -(torch.log_softmax(lg[:-1], -1).gather(1, ids[1:, None])[:, 0]).cpu().numpy(), lg.argmax(-1).cpu().numpy()


def _stable_reference(z, tag):
    """(ids [8, S], per-position NLL [8, S - 1], argmax [8, S]) of one reference run over all eight evaluated sequences"""
    ids = np.concatenate([z["ids"][None], z["ids_more"]])
    return ids, np.concatenate([z["nll_" + tag][None], z["nll_more_" + tag]]), np.concatenate([z["argmax_" + tag][None], z["argmax_more_" + tag]])


@pytest.mark.parametrize("tag", ["w8a8", "w4a8"])
def test_quantized_perplexity_within_0_05_of_the_reference_at_22_layers(dev, tag):
    """BASELINE.json: "quantized perplexity within 0.05 of reference", as a HARD bar at depth.

    tests/golden/full_depth_stable_case.npz is the reference's REAL HFForCausalLM at TinyLlama-1.1B's geometry (22 layers, hidden 2048,
    32 / 4 heads, FFN 5632) with the contractive weights of tests/seeded.py -- the embedding owns the residual stream, every branch adds
    a small correction, the unembedding is peaked (what a trained checkpoint has and the random model of full_depth_case.npz lacks) --
    under the reference's own calibration (get_act_range), surgery and mixed-precision rules, W8A8 and W4A8 (eval/harness_eval.py:75-108,
    ptq/mobilequant.py:175-201), evaluated on eight 256-token sequences (2 040 predicted tokens).  On this model the reference
    reproduces ITSELF (second run with three BLAS threads: the fixture's `*_self3` entries, asserted here to agree within 0.01), so every
    execution path of this package is held to the bar itself:

        | perplexity(path) - perplexity(reference) | <= 0.05    for the module chain, the fused prefill and the decode engine,

    plus argmax agreement >= 0.99 and, on the first sequence, logits within 0.4 % (median) / 3 % (max) of the logit span -- fixed
    numbers, about 3 x what the reference's own second run shows (0.13 % / 0.9-1.1 %).  What the bar can see: 4-bit weights move this
    model's perplexity by ~ +1.5 (thirty times the bar), 8-bit activations and weights by ~ +0.02 (W8A8 is near-lossless here, as it
    is on real checkpoints) -- a wrong or skipped WEIGHT quantizer fails the W4A8 case outright, and the logit bars catch what the W8A8
    perplexity cannot.  (Why eight sequences: over ONE sequence of 255 positions the paths sat at -0.02 (W8A8) and +0.03 ... +0.075 (W4A8)
    -- the reference's op sequence with rocBLAS instead of MKL summing the dot products at +0.03 -- because an index flip in a cached
    key / value moves every later position of that sequence the same way; tools/stable_depth_diag.py, profiles/r05.)"""
    from mobilequant_amd import llama
    from mobilequant_amd.decode import DecodeEngine
    m, z = _stable_model(dev, tag)
    ids_all, ref_nll, ref_arg = _stable_reference(z, tag)
    _, self_nll, _ = _stable_reference(z, tag + "_self3")
    fp_ppl = float(np.exp(np.concatenate([z["nll_fp"][None], z["nll_more_fp"]]).mean()))
    ref_ppl, self_ppl = float(np.exp(ref_nll.mean())), float(np.exp(self_nll.mean()))
    ref_lg = z["logits_" + tag]
    span = float(np.ptp(z["logits_fp"]))
    assert abs(self_ppl - ref_ppl) <= 0.01, ("the fixture model must be one on which the reference reproduces itself", ref_ppl, self_ppl)
    ids_t = torch.from_numpy(ids_all).long().to(dev)

    def evaluate(run):
        nlls, args, first = [], [], None
        for i in range(ids_t.shape[0]):
            lg = run(ids_t[i])
            n, a = _nll(lg, ids_t[i])
            nlls.append(n), args.append(a)
            if i == 0:
                first = lg[::8].float().cpu().numpy()
        return np.stack(nlls), np.stack(args), first
    res = {}
    with torch.no_grad():
        res["module chain"] = evaluate(lambda t: m(t.view(1, -1))[0])
        assert llama.fuse_decoder_layer(m) == 22
        res["fused prefill"] = evaluate(lambda t: m(t.view(1, -1))[0])
        eng = DecodeEngine(m, cache_len=int(ids_t.shape[1]))

        def decode(t):
            eng.reset()
            rows = []
            for tok in t.tolist():
                eng.step(tok)
                rows.append(eng.logits.clone())
            return torch.stack(rows)
        res["decode engine"] = evaluate(decode)
    report = {}
    for name, (nll, arg, sub) in res.items():
        d = np.abs(sub - ref_lg) / span
        ppl = float(np.exp(nll.mean()))
        per_seq = np.exp(nll.mean(axis=1)) - np.exp(ref_nll.mean(axis=1))          # one number per 255-position sequence
        worst = int(np.argmax(np.abs(per_seq)))
        report[name] = dict(ppl=round(ppl, 5), dppl=round(ppl - ref_ppl, 5), dppl_first_sequence=round(float(np.exp(nll[0].mean()) - np.exp(ref_nll[0].mean())), 5),
                            dppl_worst_sequence=round(float(per_seq[worst]), 5), worst_sequence=worst,
                            dppl_per_sequence=[round(float(v), 4) for v in per_seq],
                            argmax=round(float((arg == ref_arg).mean()), 4), logit_max=round(float(d.max()), 5), logit_median=round(float(np.median(d)), 6))
    print(f"stable full depth [{tag}]: {ref_nll.size} predicted tokens; fp ppl {fp_ppl:.4f}, reference {tag} ppl {ref_ppl:.4f} (quantisation moves it by "
          f"{ref_ppl - fp_ppl:+.4f}), the reference's second run {self_ppl - ref_ppl:+.5f};", report)
    for name, r in report.items():
        assert abs(r["dppl"]) <= 0.05, (tag, name, r)
        # the mean over eight sequences is the bar; a SINGLE 255-position sequence is also bounded (positions of one
        # sequence are not independent samples -- a flipped cached key / value index moves every later position the same way -- so the
        # per-sequence bound is wider: 0.12, stated; the worst sequence is printed above)
        assert abs(r["dppl_worst_sequence"]) <= 0.12, (tag, name, r)
        assert r["argmax"] >= 0.99, (tag, name, r)
        assert r["logit_median"] <= 4e-3 and r["logit_max"] <= 3e-2, (tag, name, r)


def _to_tiled(a_q):
    M, K = a_q.shape
    return a_q.view(M // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(M, K)


@pytest.mark.parametrize("spread", [40.0, 12.0])
def test_fused_epilogue_quantizer_flip_rate_against_the_reference_divide_form_at_the_headline_shape(dev, spread):
    """The GEMM kernels form an 8-bit output index with ONE fma on pre-divided constants,
        index = sat_u8(rne(float(t) * (alpha / s_o) + (bias / s_o + o))),
    where the reference's output quantizer divides (qmodule.py:286-287 on qmodule.py:353's fp32 value):
        y = float(t) * alpha + bias;  index = clamp(rint(y / s_o) + o, 0, 255).
    A deliberate deviation, so far quantified only by tools/epilogue_flip_rate.py.  Here, at the headline shape
    (2048 x 2048 -> 5632) on quantised-Gaussian operands with per-channel weight scales and a bias: from torch._int_mm's EXACT int32
    accumulators both forms are evaluated in fp32; the kernel must (a) be its own formula on every output and (b) differ from the
    reference's divide form on at most 1e-5 of the 11.5 M outputs, never by more than one step."""
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    M, K, N = 2048, 2048, 5632
    g = torch.Generator(device="cpu").manual_seed(3)
    a = (torch.randn(M, K, generator=g) * spread).round().clamp(-128, 127).to(torch.int8).to(dev)
    w = (torch.randn(N, K, generator=g) * spread).round().clamp(-128, 127).to(torch.int8).to(dev)
    sa, sw = 0.031, (torch.rand(N, generator=g) * 4e-3 + 1e-3).to(dev)
    alpha = (sa * sw).float()
    bias = (torch.randn(N, generator=g) * 0.2).to(dev)
    t = torch._int_mm(a, w.t().contiguous())                   # exact int32 accumulators (zero points 0: t is the whole integer part)
    y = t.float() * alpha + bias
    sample = y.flatten()[::97].float()
    lo, hi = torch.quantile(sample, 0.001), torch.quantile(sample, 0.999)
    so = ((hi - lo) / 255).reshape(1)
    oo = torch.round(-lo / so).reshape(1)
    zero = torch.zeros(N, dtype=torch.int32, device=dev)
    rs = a.to(torch.int32).sum(1, dtype=torch.int32)
    got = ops.int8_linear(_to_tiled(a), w, rs, alpha, zero, zero, bias, out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=255.0,
                          out_dtype=MQ_U8, a_tiled_rows=M).float()
    inv = (1.0 / so).float()
    # the fma in float64: the product of two fp32 is exact there, the sum rounds once more before the fp32 conversion (double rounding:
    # a handful of ties in 11.5 M at most)
    own = torch.clamp(torch.round((t.double() * (alpha * inv).double() + (bias * inv + oo).double()).float()), 0, 255)
    ref = torch.clamp(torch.round(y / so) + oo, 0, 255)
    d = (got - ref).abs()
    rate = float((d != 0).double().mean())
    print(f"epilogue flip rate vs the divide form (operand sigma {spread}): {int((d != 0).sum())} of {d.numel()} = {rate:.2e}, max {int(d.max())} LSB; "
          f"saturated {float(((ref == 0) | (ref == 255)).double().mean()):.3%}")
    assert int((got != own).sum()) <= 4, ("the kernel is not its documented one-fma formula", int((got != own).sum()))
    assert int(d.max()) <= 1 and rate <= 1e-5, (rate, int(d.max()))


# ---- the standalone integer QMatMul (qmodule.py:453-466) --------------------------------------------------------
def _grid(bits, sym, lo, hi):
    from oracle import mq_oracle as O
    g = O.QuantizerOracle(bits, is_symmetric=sym)
    g.set_from_minmax(np.float32(lo), np.float32(hi))
    return g


def _dev_grid(g, dev):
    return (torch.tensor([float(g.scale)], device=dev), torch.tensor([float(g.offset)], device=dev), float(g.qmin), float(g.qmax))


QMM_CASES = [
    # lead, M, N, K, x2 given as k-contiguous view, (bits, symmetric) of x1, of x2, of the output (None = no quantizer)
    ((2, 3), 77, 100, 36, True, (8, False), (8, False), (16, False)),         # qk_bmm, ragged everything (N % 4 need not hold: k^T view)
    ((1, 4), 128, 192, 64, True, (8, False), (8, False), (16, False)),        # qk_bmm at head_dim 64
    ((2, 2), 50, 64, 200, False, (16, False), (8, False), (8, False)),        # pv_bmm: 16-bit probabilities, K ragged against the 64-chunk
    ((1, 3), 130, 128, 1024, False, (16, False), (8, False), (8, False)),     # pv_bmm, long K (integer sums beyond 2^24)
    ((3,), 65, 36, 260, False, (8, True), (8, True), (8, True)),              # symmetric (signed) grids
    ((1,), 16, 8, 4, False, (8, False), (8, False), None),                    # tiny, no output quantizer
    ((2,), 200, 72, 128, True, (4, False), (8, True), (16, True)),            # 4-bit x1, signed 16-bit output
    ((1, 2), 64, 256, 256, True, (12, True), (6, False), (8, False)),         # 12-bit signed x1 on the two-plane path, 6-bit x2
    ((), 1, 512, 64, True, (8, False), (8, False), (16, False)),              # a decode step's q.k^T: M = 1, no leading dim
    ((2,), 1, 64, 9, False, (16, False), (8, False), (8, False)),             # a decode step's p.v over 9 cached positions: K % 4 != 0
    ((3,), 5, 7, 13, True, (8, False), (8, True), (8, False)),                # nothing is a multiple of anything
    ((1,), 40, 10, 66, False, (8, False), (8, False), (16, False)),           # dense x2 with N % 4 != 0: handed over K-contiguous
]


@pytest.mark.parametrize("case", QMM_CASES, ids=[f"{c[1]}x{c[2]}x{c[3]}_{'kT' if c[4] else 'kn'}_{c[5][0]}b{c[6][0]}b" for c in QMM_CASES])
def test_integer_qmatmul_every_output_against_the_exact_integer_oracle(dev, case):
    """ops.qmatmul (mq_qmatmul: both input quantizers, an exact int8 MFMA contraction and the output quantizer in one launch) against
    oracle.qmatmul_exact -- the reference's QMatMul.forward (qmodule.py:453-466) with its contraction carried out exactly over the
    quantizer indices -- on EVERY output, bit for bit: arbitrary M / N / K (no multiple-of-64 or mask assumption), both memory orders
    of x2, 4- ... 16-bit unsigned and signed grids, ranges that do not straddle zero symmetrically."""
    from mobilequant_amd import ops
    from oracle import mq_oracle as O
    lead, M, N, K, kt, (b1, s1), (b2, s2), bo = case
    rng = np.random.default_rng(1000 * M + N + K)
    if b1 > 8 and not s1:       # probabilities
        a = rng.random(lead + (M, K), dtype=np.float32) ** 4
        a /= a.sum(-1, keepdims=True)
        g1 = _grid(b1, s1, 0.0, float(a.max()))
    else:
        a = (rng.standard_normal(lead + (M, K), dtype=np.float32) * 1.3 + 0.2).astype(np.float32)
        g1 = _grid(b1, s1, float(a.min()) * 0.9, float(a.max()) * 0.9)            # some elements clamp
    b = (rng.standard_normal(lead + (K, N), dtype=np.float32) * 0.8 - 0.1).astype(np.float32)
    g2 = _grid(b2, s2, float(b.min()) * 0.95, float(b.max()) * 0.95)
    fp = np.matmul(a, b)
    go = None if bo is None else _grid(bo[0], bo[1], float(np.percentile(fp, 0.5)), float(np.percentile(fp, 99.5)))
    want = O.qmatmul_exact(a, b, g1, g2, go)
    ta = torch.from_numpy(a).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(np.swapaxes(b, -1, -2))).to(dev).transpose(-1, -2) if kt else torch.from_numpy(b).to(dev)
    got = ops.qmatmul(ta, tb, _dev_grid(g1, dev), _dev_grid(g2, dev), None if go is None else _dev_grid(go, dev)).cpu().numpy()
    assert got.shape == want.shape
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), (case, int(bad.sum()), np.argwhere(bad)[:5].tolist(), got[bad][:5].tolist(), want[bad][:5].tolist())
    if go is not None:          # the output grid is exercised, not saturated
        idx = np.rint(want / go.scale + go.offset)
        assert np.unique(idx).size > min(64, (go.qmax - go.qmin) / 4)


def test_qmatmul_module_takes_the_integer_kernel_and_matches_its_own_simulated_path(dev):
    """QMatMul.forward routes static per-tensor grids to mq_qmatmul under no_grad (int8_coverage lists it), tags its output with the
    output grid, keeps the simulated path (HIP fake-quant kernels around the library matmul) for what the kernel does not serve -- a
    gradient, a dynamic grid, broadcasting operands -- and the two paths agree within one output step."""
    import mobilequant_amd as mq
    from mobilequant_amd.quantization import qmodule as Q
    torch.manual_seed(3)
    q = torch.randn(2, 4, 70, 64, device=dev)
    k = torch.randn(2, 4, 90, 64, device=dev)
    mod = mq.QMatMul(mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16))
    fp = torch.matmul(q, k.transpose(2, 3))
    mod.set_scale_offset({"input": [float(q.min()), float(q.max())], "input2": [float(k.min()), float(k.max())],
                          "output": [float(fp.min()), float(fp.max())]}, "buffer")
    holder = torch.nn.Module()
    holder.qk_bmm = mod
    with torch.no_grad():
        y = mod(q, k.transpose(2, 3))
        assert Q._producer_grid(y) is mod.output_quantizer
        cov = mq.int8_coverage(holder, reset=True)
        assert cov["int8_calls"] == 1 and cov["simulated_calls"] == 0, cov["summary"]
        mod.int8_mode = "off"
        y_sim = mod(q, k.transpose(2, 3))
        mod.int8_mode = "auto"
        mq.int8_coverage(holder, reset=True)
    d = (y - y_sim).abs()
    lsb = float(mod.output_quantizer.scale)
    assert float(d.max()) <= lsb * 1.001 and float((d == 0).float().mean()) > 0.99
    # what stays simulated, and says why
    qg = q.clone().requires_grad_(True)
    mod(qg, k.transpose(2, 3)).sum().backward()
    assert qg.grad is not None
    with torch.no_grad():
        mod(q[:, :1], k.transpose(2, 3))                       # broadcasting leading dims: torch.matmul's job
    cov = mq.int8_coverage(holder)
    assert cov["simulated_calls"] == 2 and any("gradient" in k_ for k_ in cov["modules"]["qk_bmm"]), cov["summary"]


@pytest.mark.parametrize("tag", ["w4", "stablelm", "gemma"])
def test_no_qmatmul_of_the_three_model_families_runs_on_the_library_bmm(dev, tag):
    """VERDICT r04 item 4's acceptance: on the module CHAIN (no fused attention) of the three families of BASELINE.json -- llama leaf
    graph, StableLM-2 (LayerNorm, partial rotary, head_dim 64 MHA), Gemma (head_dim 256, MQA) -- int8_coverage() lists every qk_bmm /
    pv_bmm on the integer kernel, none on the simulated path, for a prefill forward and for a cached decode step (M = 1); and the
    chain still reproduces the reference's logits (the decode_case_* fixtures) within the bars of the recipes test."""
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd import llama
    from seeded import seeded_parameters_
    from test_llama_host import FAMILY_SHAPES
    z = load_npz(f"decode_case_{tag}.npz")
    shape_kw = FAMILY_SHAPES.get(tag) or dict(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=96, eps=1e-5, max_pos=64,
                                              hidden_act="silu")
    m = llama.LlamaForCausalLM(llama.LlamaShape(**shape_kw)).eval()
    seeded_parameters_(m, std=0.08)
    m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}      # noqa: E731
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    mq.update_qcfg(m, strip(json.loads(str(z["qcfg"]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer")
    mq.wire_integer_inputs(m)
    m.requires_grad_(False)
    ids = torch.from_numpy(z["ids"]).long()
    ref, span = z["logits_w4a8"][0], float(np.ptp(z["logits_fp"]))
    mq.int8_coverage(m, reset=True)
    with torch.no_grad():
        chain = m(ids[None].to(dev))[0].cpu().numpy()
    cov = mq.int8_coverage(m, reset=True)
    bmm = {n: c for n, c in cov["modules"].items() if n.endswith("_bmm")}
    assert len(bmm) == 2 * len(m.layers) and all(list(c) == ["int8"] for c in bmm.values()), bmm
    d = np.abs(chain - ref) / span
    assert d.max() <= 0.045 and np.quantile(d, 0.99) <= 0.021 and np.median(d) <= 1e-5, (float(d.max()), float(np.median(d)))
    # a cached decode step of the module graph: q is one row, k / v come from the static cache
    cache = m.new_cache(1, 64) if hasattr(m, "new_cache") else None
    if cache is not None:
        with torch.no_grad():
            m(ids[None, :8].to(dev), cache=cache, pos=0)
            m(ids[None, 8:9].to(dev), cache=cache, pos=8)
        cov = mq.int8_coverage(m, reset=True)
        assert all(list(c) == ["int8"] for n, c in cov["modules"].items() if n.endswith("_bmm")), cov["summary"]


# ---- calibration: statistics where the tensors are produced -----------------------------------------------------
@pytest.mark.parametrize("H,S,T,masked", [(4, 96, 96, True), (3, 130, 2048, False), (2, 64, 4096, True), (1, 1, 36, False), (2, 257, 512, True)])
def test_calibration_score_chain_is_the_torch_chain_and_its_statistics_are_exact(dev, H, S, T, masked):
    """mq_calib_attention_probs against the graph it replaces (hf_model.py:513-530 between generate_act_range.py:55-69's two hooks):
    probabilities within 8 ulp / 1.2e-7 of torch's `softmax(raw / sqrt(d) + mask)` (another exp and another summation order than torch's
    softmax kernel), written in place; the raw-score statistic is EXACTLY
    torch's min / max of the scores, the probability statistic exactly the min / max of the probabilities the kernel wrote (and within
    2 ulp of torch's); running statistics accumulate over calls; a NaN score makes both statistics NaN, as amin / amax do."""
    from mobilequant_amd import ops
    torch.manual_seed(S + T)
    raw = torch.randn(1, H, S, T, device=dev) * 3.0
    mask = None
    if masked:
        mask = torch.full((S, T), float("-inf"), device=dev).triu(T - S + 1)
    sqrt_d = 8.0 if T != 2048 else float(np.sqrt(128.0))
    want = torch.softmax(raw / sqrt_d + (mask if mask is not None else 0.0), dim=-1, dtype=torch.float32)
    stats = [ops.minmax_new(1, dev) for _ in range(2)]
    keep = raw.clone()
    got = ops.calib_attention_probs_(raw, mask, sqrt_d, stats[0][0], stats[0][1], stats[1][0], stats[1][1])
    assert got.data_ptr() == raw.data_ptr()
    ulp = torch.abs(got.view(torch.int32) - want.view(torch.int32))
    assert int(ulp.max()) <= 8 and float((got - want).abs().max()) <= 1.2e-7, (int(ulp.max()), float((got - want).abs().max()))
    assert float(stats[0][0]) == float(keep.min()) and float(stats[0][1]) == float(keep.max())
    assert float(stats[1][0]) == float(got.min()) and float(stats[1][1]) == float(got.max())
    assert abs(float(stats[1][1]) - float(want.max())) <= 3e-7 * float(want.max())
    # running: a second tensor can only widen them
    raw2 = keep * 0.5
    ops.calib_attention_probs_(raw2, mask, sqrt_d, stats[0][0], stats[0][1], stats[1][0], stats[1][1])
    assert float(stats[0][0]) == float(keep.min()) and float(stats[0][1]) == float(keep.max())
    raw3 = keep.clone()
    raw3[0, 0, 0, 0] = float("nan")
    ops.calib_attention_probs_(raw3, mask, sqrt_d, stats[0][0], stats[0][1], stats[1][0], stats[1][1])
    assert all(np.isnan(float(t)) for pair in stats for t in pair)


@pytest.mark.parametrize("family", ["tinyllama", "stablelm", "gemma"])
def test_get_act_range_with_fused_attention_statistics_equals_the_hook_path(dev, family):
    """get_act_range on this package's leaf graphs takes qk_bmm.output / pv_bmm.input from the fused score chain (llama.Attention asks
    the attached ActRangeCollector): everything in front of the first softmax (layer 0's norm, q / k / v, qk_bmm incl. the raw-score statistic)
    equals the all-hooks run exactly; what depends on the probabilities downstream agrees to summation-order tolerance (the fused
    chain's probabilities sit a few ulp from torch's softmax kernel); the hooks re-read two thirds fewer bytes."""
    from mobilequant_amd.calibration import ActRangeCollector, get_act_range
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from test_llama_host import FAMILY_SHAPES
    kw = FAMILY_SHAPES.get(family) or dict(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=96, eps=1e-5, max_pos=128)
    kw = dict(kw, max_pos=128)
    m = LlamaForCausalLM(LlamaShape(**kw))
    m.reset_parameters(seed=4, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(9)
    samples = [torch.randint(0, kw["vocab"], (1, 100), generator=g) for _ in range(3)]
    fused = get_act_range(m, samples)
    ActRangeCollector.fuse_attention_statistics = False
    try:
        hooks = get_act_range(m, samples)
    finally:
        ActRangeCollector.fuse_attention_statistics = True
    assert fused.keys() == hooks.keys()
    exact = 0
    for name in hooks:
        for field, (lo, hi) in hooks[name].items():
            flo, fhi = fused[name][field]
            if name.startswith("layers.0.") and (name.endswith("qk_bmm") or "input_layernorm" in name or name.endswith(("q_proj", "k_proj", "v_proj"))
                                                 or (name.endswith("pv_bmm") and field == "input2")):
                assert (flo, fhi) == (lo, hi), (name, field)          # everything in front of the first softmax is the same arithmetic
                exact += 1
            else:
                tol = 2e-5 * max(abs(lo), abs(hi), 1e-6)
                assert abs(flo - lo) <= tol and abs(fhi - hi) <= tol, (name, field, (flo, fhi), (lo, hi))
    assert exact >= 12
    assert any(n.endswith("pv_bmm") and fused[n]["input"][0] == 0.0 and 0.0 < fused[n]["input"][1] <= 1.0 for n in fused)


# ---- ADVICE r04 --------------------------------------------------------------------------------------------------------------------
def _small_sim_llama(dev, seed=6):
    import mobilequant_amd as mq
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    from toy_models import apply_mixed_precision
    m = LlamaForCausalLM(LlamaShape(hidden=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=64, max_pos=128))
    m.reset_parameters(seed=seed, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 64, (1, 96), generator=g)
    act = get_act_range(m, [ids, torch.randint(0, 64, (1, 96), generator=g)])
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8))
    apply_mixed_precision(m, mq)
    return m, act, ids.to(dev)


def test_per_group_o_proj_under_the_fused_decoder_layer_falls_through_to_its_own_kernel(dev):
    """the fused attention asked o_proj for its integer weight plan BEFORE asking whether o_proj is on the integer
    path at all -- a per-group weight grid (round 4's mq_w8a8_linear_grouped recipe) then raised (grid already loaded) or registered a
    per-ROW grid on a per-group quantizer (first forward).  Now the guard comes first: under fuse_decoder_layer a per-group o_proj is
    handed the attention output as a tensor and runs its own grouped kernel -- with no weight grid yet (first forward derives the
    per-group grid: [N * G] scales) and again with the grid in place -- and the layer agrees with the unfused module chain."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    m, act, ids = _small_sim_llama(dev)
    for name, mod in m.named_modules():
        if name.endswith("o_proj"):
            mod.weight_quantizer.qcfg.is_per_channel, mod.weight_quantizer.qcfg.group_size = True, 64
    mq.set_scale_and_offset(m, act, "buffer")
    mq.wire_integer_inputs(m)
    with torch.no_grad():
        chain = m(ids)[0]
    o0 = m.layers[0].self_attn.o_proj
    assert o0.weight_quantizer.scale.numel() == 256 * (256 // 64)          # a per-GROUP grid came out of the first forward
    for mod in m.modules():                                                   # start again without weight grids: the fused pass sees none
        if isinstance(mod, mq.QLinear) and hasattr(mod.weight_quantizer, "scale"):
            del mod.weight_quantizer.scale, mod.weight_quantizer.offset
    mq.int8_coverage(m, reset=True)
    with torch.no_grad():
        assert llama.fuse_decoder_layer(m) == 2
        first = m(ids)[0]
        assert o0.weight_quantizer.scale.numel() == 256 * (256 // 64)
        second = m(ids)[0]
    assert torch.equal(first, second)
    cov = mq.int8_coverage(m)
    assert cov["simulated_calls"] == 0, cov["summary"]
    span = float(chain.max() - chain.min())
    d = (first - chain).abs() / span
    assert float(d.max()) <= 0.05 and float(d.median()) <= 1e-3, (float(d.max()), float(d.median()))


def test_fused_gated_mlp_leaves_dynamic_own_input_quantizers_to_the_module_chain(dev):
    """w1 / w3 carrying their OWN dynamic input quantizers have no grid before the first refresh (and a stale one
    after it); the fused FFN block used to read g1.grid_token() there -- AttributeError on the first forward.  It now stays on the
    module chain, whose per-call refresh is the reference's `is_dynamic` behaviour (qmodule.py:262-277)."""
    import mobilequant_amd as mq
    m, act, ids = _small_sim_llama(dev, seed=8)
    mq.set_scale_and_offset(m, act, "buffer")
    mq.wire_integer_inputs(m)
    dyn = mq.QuantConfig(bitwidth=8, is_dynamic=True)
    for layer in m.layers:
        for lin in (layer.mlp.w1, layer.mlp.w3):
            lin.input_quantizer = mq.Quantizer(dyn)
    with torch.no_grad():
        chain = m(ids)[0]
        assert mq.fuse_gated_mlp(m) == 2
        for lin in (m.layers[0].mlp.w1, m.layers[0].mlp.w3):             # back to "never ran": no scale attribute at all
            for attr in ("scale", "offset"):
                if hasattr(lin.input_quantizer, attr):
                    delattr(lin.input_quantizer, attr)
        fused = m(ids)[0]
    assert torch.equal(fused, chain)


@pytest.mark.parametrize("S,heads,kv_heads,rot,chunks,p_top", [(2048, 8, 2, 64, 1, 1.0), (704, 4, 1, 64, 1, 1.0), (130, 2, 2, 64, 1, 1.0), (320, 4, 4, 16, 1, 1.0),
                                                              (448, 4, 2, 64, (128, 256, 64), 1.0), (704, 4, 2, 64, 1, 0.4), (192, 2, 1, 16, 1, 0.05)])
def test_attention_f16_score_contraction_is_the_int8_one_bit_for_bit(dev, S, heads, kv_heads, rot, chunks, p_top):
    """head_dim 64 with a 16-bit score grid: the scores contracted as v_mfma_f32_16x16x32_f16 over fp16 images of the CENTRED indices
    (K / vT tiles staged once per workgroup in an LDS ring, software-pipelined) against the int8 MFMA + zero-point terms
    (mq_attention_set_f16(0)): sum (qi - zq)(ki - zk) < 2^24 is exact in the fp32 accumulator, so fp32 output, int8 image and row
    sums must be equal bit for bit -- full and partial rotary (the q image from the prep kernel), every mix of recomputed / parked key
    blocks (S = 2048: 32 key blocks), ragged S, a chunked prefill over the cached images (the fp16 K cache), and a probability grid that
    does NOT hold [0, 1] (p_top < 1: the f16 form then keeps the index clamp it otherwise drops as dead)."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    from test_gpu_round2 import _grid_of
    from test_gpu_round3 import _case
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, 64, rot, seed=S + rot)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    if p_top != 1.0:
        from test_gpu_round3 import _mk
        grids["pv_a"] = _grid_of(_mk(16, 0.0, p_top), dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                       # noqa: E731
    outs = []
    for f16 in (0, 1):
        L.load().mq_attention_set_f16(f16)
        try:
            img = torch.zeros(S, heads * 64, dtype=torch.int8, device=dev)
            rs = torch.zeros(S, dtype=torch.int32, device=dev)
            if chunks == 1:
                out = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, image=(img, rs, 0, 128, False))
            else:
                cache = ops.attention_image_cache(kv_heads, 64, S, dev)
                parts, a0 = [], 0
                for n in chunks:
                    parts.append(ops.attention_quant(t(q[a0:a0 + n]), t(k[a0:a0 + n]), t(v[a0:a0 + n]), t(cos[a0:a0 + n]), t(sin[a0:a0 + n]), heads, kv_heads,
                                                     grids, image=(img, rs, a0, 128, False), cache=cache, pos0=a0))
                    a0 += n
                assert a0 == S
                out = torch.cat(parts)
            torch.cuda.synchronize()
            outs.append((out, img, rs))
        finally:
            L.load().mq_attention_set_f16(1)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    if chunks == 1:        # ... and the q rows prepared inside the attention workgroups (full rotary, and rot_dim 16) == the prep kernel's q image
        L.load().mq_attention_set_fused_q(0)
        try:
            img = torch.zeros(S, heads * 64, dtype=torch.int8, device=dev)
            rs = torch.zeros(S, dtype=torch.int32, device=dev)
            out = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, image=(img, rs, 0, 128, False))
            torch.cuda.synchronize()
        finally:
            L.load().mq_attention_set_fused_q(1)
        assert torch.equal(out, outs[1][0]) and torch.equal(img, outs[1][1]) and torch.equal(rs, outs[1][2])


@pytest.mark.parametrize("B,S,heads,kv_heads,D,rot", [(3, 192, 4, 2, 64, 64), (2, 100, 2, 1, 64, 64), (2, 130, 4, 4, 64, 16), (2, 128, 2, 1, 256, 256), (3, 70, 2, 2, 128, 128)])
def test_attention_batch_in_one_launch_equals_one_call_per_sequence(dev, B, S, heads, kv_heads, D, rot):
    """mq_attention_args.batch: B sequences in ONE prep + core launch pair (grid z = sequence; inputs, outputs and scratch laid out
    [B][...]) give, bit for bit, what B single-sequence calls give -- fp32 output, the int8 image for o_proj (sequence b on rows
    b * S ...) and its row sums; ragged S (padded per sequence), partial rotary, head_dim 64 / 128 / 256, fp32 and index inputs."""
    from mobilequant_amd import ops
    from test_gpu_round2 import _grid_of
    from test_gpu_round3 import _case
    cases = [_case(S, heads, kv_heads, D, rot, seed=17 * b + S) for b in range(B)]
    qk, pv = cases[0][5], cases[0][6]                                # one set of grids for the batch
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                       # noqa: E731
    q, k, v = (torch.stack([t(c[i]) for c in cases]) for i in (0, 1, 2))
    cos, sin = t(cases[0][3]), t(cases[0][4])
    img1, rs1 = torch.zeros(B * S, heads * D, dtype=torch.int8, device=dev), torch.zeros(B * S, dtype=torch.int32, device=dev)
    img2, rs2 = torch.zeros_like(img1), torch.zeros_like(rs1)
    one = torch.stack([ops.attention_quant(q[b], k[b], v[b], cos, sin, heads, kv_heads, grids, image=(img1, rs1, b * S, 128, False), head_dim=D) for b in range(B)])
    all_ = ops.attention_quant(q, k, v, cos, sin, heads, kv_heads, grids, image=(img2, rs2, 0, 128, False), head_dim=D)
    torch.cuda.synchronize()
    assert all_.shape == one.shape and torch.equal(all_, one) and torch.equal(img1, img2) and torch.equal(rs1, rs2)
    # index input (the fused q|k|v GEMM's uint8 output) through the same two routes
    g = torch.Generator(device="cpu").manual_seed(S)
    idx = torch.randint(0, 256, (B, S, (heads + 2 * kv_heads) * D), dtype=torch.uint8, generator=g).to(dev)
    ig = tuple((torch.tensor([0.05], device=dev), torch.tensor([float(z)], device=dev)) for z in (128.0, 120.0, 131.0))
    one = torch.stack([ops.attention_quant(None, None, None, cos, sin, heads, kv_heads, grids, qkv_idx=(idx[b], ig), head_dim=D) for b in range(B)])
    all_ = ops.attention_quant(None, None, None, cos, sin, heads, kv_heads, grids, qkv_idx=(idx, ig), head_dim=D)
    torch.cuda.synchronize()
    assert torch.equal(all_, one)


def test_fused_model_forward_of_a_batch_is_the_forward_of_each_sequence(dev):
    """The fused decoder layer on ids [3, 160]: the attention of the whole batch is ONE launch pair (no per-sequence Python loop), the
    linears see 3 x 160 rows; every row's quantized computation is the same as in a forward of its sequence alone (2-layer W8A8 model
    of decode_case.npz; ragged S: 160 = 2.5 key blocks)."""
    import dataclasses
    from test_gpu_round2 import _decode_model
    from mobilequant_amd import llama, ops
    m, z = _decode_model(dev)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=256))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(3, m.shape.vocab, (3, 160), generator=g).to(dev)
    calls = []
    real = ops.attention_quant
    ops.attention_quant = lambda *a, **k: (calls.append(a[0].dim() if a[0] is not None else k["qkv_idx"][0].dim()), real(*a, **k))[1]
    try:
        with torch.no_grad():
            assert llama.fuse_decoder_layer(m) == 2
            whole = m(ids)
            n_batched = len(calls)
            each = torch.cat([m(ids[b:b + 1]) for b in range(3)])
    finally:
        ops.attention_quant = real
    torch.cuda.synchronize()
    assert n_batched == 2 and calls[:2] == [3, 3]                  # one call per layer, 3-D (batched) input
    # (the fp32 lm_head is a library GEMM whose kernel -- and summation order -- depends on the row count: equal up to that; one
    # flipped 8-bit index anywhere upstream would move a logit by ~1e-2)
    assert float((whole - each).abs().max()) <= 1e-4


def test_last_logits_only_is_the_last_row_of_the_full_forward(dev):
    """LlamaForCausalLM.forward(last_logits_only=True) -- what DecodeEngine.prefill runs -- computes final norm + lm_head on the last
    position only: the same decoder stack, so its logits are the full forward's last row up to the library GEMM's summation order."""
    import dataclasses
    from test_gpu_round2 import _decode_model
    from mobilequant_amd import llama
    m, z = _decode_model(dev)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=256))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    ids = torch.randint(3, m.shape.vocab, (2, 130), generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        assert llama.fuse_decoder_layer(m) == 2
        full = m(ids)
        last = m(ids, last_logits_only=True)
    torch.cuda.synchronize()
    assert last.shape == (2, 1, m.shape.vocab)
    assert float((last[:, 0] - full[:, -1]).abs().max()) <= 1e-4
