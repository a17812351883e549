"""Round-3 GPU parity tests (through the C ABI): the decode step's attention launch over the integer KV cache against the numpy
restatement of the reference's attention, position by position; split-position launches against the single-workgroup one."""
import ctypes

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


def _mk(bits, lo, hi):
    o = O.QuantizerOracle(bitwidth=bits)
    o.set_from_minmax(F32(lo), F32(hi))
    return o


def _case(S, heads, kv_heads, D, rot, seed, qk_out_bits=16, pv_out_bits=8):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((S, heads * D), dtype=F32) * 1.5
    k = rng.standard_normal((S, kv_heads * D), dtype=F32) * 1.5
    v = rng.standard_normal((S, kv_heads * D), dtype=F32)
    inv = 1.0 / (10000.0 ** (np.arange(0, rot, 2, dtype=F32) / rot))
    ang = np.outer(np.arange(S, dtype=F32), inv).astype(F32)
    ang = np.concatenate((ang, ang), axis=-1)
    cos, sin = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    qr = O.rope_partial(q.reshape(S, heads, D).transpose(1, 0, 2), cos, sin)
    kr = O.rope_partial(k.reshape(S, kv_heads, D).transpose(1, 0, 2), cos, sin)
    sc = np.einsum("hsd,htd->hst", qr, np.repeat(kr, heads // kv_heads, axis=0))
    qk = (_mk(8, qr.min(), qr.max()), _mk(8, kr.min(), kr.max()), _mk(qk_out_bits, sc.min(), sc.max()) if qk_out_bits else None)
    pv = (_mk(16, 0.0, 1.0), _mk(8, v.min(), v.max()), _mk(pv_out_bits, -0.8 * np.abs(v).max(), 0.8 * np.abs(v).max()) if pv_out_bits else None)
    return q, k, v, cos, sin, qk, pv


class _Stepper:
    """mq_decode_attention driven directly: one launch per position over its own int8 cache."""

    def __init__(self, dev, heads, kv_heads, D, rot, cache_len, cos, sin, qk, pv, o_in, nsplit):
        from mobilequant_amd import _lib
        from mobilequant_amd._lib import MqDecodeAttentionArgs, MqGrid
        self.lib, self.dev = _lib, dev
        self.keep = []

        def grid(o):
            if o is None:
                return MqGrid(None, None, 0.0, 0.0)
            s, z = torch.tensor([float(o.scale)], device=dev), torch.tensor([float(o.offset)], device=dev)
            self.keep += [s, z]
            return MqGrid(s.data_ptr(), z.data_ptr(), float(o.qmin), float(o.qmax))
        a = MqDecodeAttentionArgs()
        self.qkv = torch.zeros((heads + 2 * kv_heads) * D, device=dev)
        self.kc = torch.zeros(kv_heads, cache_len, D, dtype=torch.int8, device=dev)
        self.vc = torch.zeros(kv_heads, cache_len, D, dtype=torch.int8, device=dev)
        self.cos, self.sin = torch.from_numpy(cos).to(dev).contiguous(), torch.from_numpy(sin).to(dev).contiguous()
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.out = torch.zeros(heads * D, device=dev)
        self.out_q = torch.zeros(heads * D, dtype=torch.int8, device=dev)
        self.part = torch.zeros(nsplit, heads * D, dtype=torch.int64, device=dev)
        self.ticket = torch.zeros(heads, dtype=torch.int32, device=dev)
        a.qkv, a.k_cache, a.v_cache, a.cos, a.sin, a.pos = (t.data_ptr() for t in (self.qkv, self.kc, self.vc, self.cos, self.sin, self.pos))
        a.heads, a.kv_heads, a.head_dim, a.cache_len, a.rot_dim, a.nsplit = heads, kv_heads, D, cache_len, rot, nsplit
        a.qk_a, a.qk_b, a.qk_out, a.pv_a, a.pv_b, a.pv_out, a.o_in = (grid(o) for o in (*qk, *pv, o_in))
        grids = (MqGrid * 7)(a.qk_a, a.qk_b, a.qk_out, a.pv_a, a.pv_b, a.pv_out, a.o_in)
        self.consts = torch.zeros(64, device=dev)
        _lib.call("mq_decode_pack_grids", grids, 7, self.consts.data_ptr(), torch.cuda.current_stream().cuda_stream)
        a.consts, a.out, a.out_q, a.part, a.ticket = (t.data_ptr() for t in (self.consts, self.out, self.out_q, self.part, self.ticket))
        self.a = a

    def step(self, t, q_row, k_row, v_row):
        self.qkv.copy_(torch.from_numpy(np.concatenate((q_row, k_row, v_row))).to(self.dev))
        self.pos.fill_(int(t))
        self.lib.call("mq_decode_attention", ctypes.byref(self.a), torch.cuda.current_stream().cuda_stream)
        return self.out.cpu().numpy().copy(), self.out_q.cpu().numpy().copy()


@pytest.mark.parametrize("S,heads,kv_heads,D,rot,qk_out_bits,pv_out_bits", [
    (70, 4, 2, 64, 64, 16, 8), (40, 2, 1, 256, 256, 16, 8), (48, 4, 4, 64, 16, 16, 8), (36, 2, 2, 128, 128, 16, 8), (40, 8, 2, 32, 32, 0, 0),
    (200, 2, 1, 64, 64, 16, 16)])
def test_decode_attention_every_position_vs_oracle(dev, S, heads, kv_heads, D, rot, qk_out_bits, pv_out_bits):
    """mq_decode_attention fed a sequence ONE position at a time (RoPE incl. partial rotary, cache append, exact integer q.k and
    p.v, fp32 softmax, exact-form quantizers) must reproduce row t of the reference's causal attention (oracle.attention_sim:
    hf_model.py:486-534 + qmodule.py:453-466) for every t: the contractions are exact where the reference's fp32 matmuls round, so
    an output may sit one step of its 8-bit grid off on a small fraction of elements.  The int8 image for o_proj is the index of the
    fp32 output on the consumer's grid; launches that split the cached positions over 3 / 4 workgroups per head are bit-identical
    to the single-workgroup launch (the partial sums are integers); one step past the cache does nothing."""
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, D, rot, seed=S + D + heads, qk_out_bits=qk_out_bits, pv_out_bits=pv_out_bits)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    o_in = pv[2] if (pv[2] is not None and pv[2].qmax == 255) else _mk(8, want.min(), want.max())
    runs = {}
    for nsplit in (1, 3, 4):
        st = _Stepper(dev, heads, kv_heads, D, rot, S, cos, sin, qk, pv, o_in, nsplit)
        outs = [st.step(t, q[t], k[t], v[t]) for t in range(S)]
        runs[nsplit] = (np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]))
        before = (st.out.clone(), st.kc.clone())
        st.step(S, q[0], k[0], v[0])                                   # position == cache_len
        assert torch.equal(st.out, before[0]) and torch.equal(st.kc, before[1])
        assert int(st.ticket.abs().sum()) == 0                          # the tickets reset themselves
    got, got_q = runs[1]
    for nsplit in (3, 4):
        assert np.array_equal(runs[nsplit][0].view(np.uint32), got.view(np.uint32)) and np.array_equal(runs[nsplit][1], got_q)
    assert got.shape == want.shape and np.isfinite(got).all()
    diff = np.abs(got - want)
    span = float(want.max() - want.min())
    if pv_out_bits == 8:
        step = float(pv[2].scale)
        assert diff.max() <= 1.001 * step, (diff.max(), step)
        assert (diff > 0.5 * step).mean() < 0.02, (diff > 0.5 * step).mean()
    else:
        assert diff.max() <= 2e-3 * span, (diff.max(), span)
    assert np.median(diff) <= 2e-4 * span
    # the int8 image: index of the fp32 output on the consumer's grid (oracle quantizer), storage index - 128
    idx = o_in.forward(got.astype(F32), return_index=True)[1].astype(np.int32) - 128
    assert np.array_equal(idx, got_q.astype(np.int32))


def test_channel_scale_set_after_a_forward_requantises_the_weight_on_its_new_range(dev):
    """QLinear.set_input_channel_scale AFTER the linear already ran must not keep the weight grid derived from W: the
    reference folds first and quantises after (ptq/smoothquant.py:64-69), so the integer weights of W * s sit on the range of W * s --
    the same numbers as a fresh module whose weight is W * s.  A calibrated / loaded weight grid is left alone."""
    import mobilequant_amd as mq
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(96, 128, generator=g) * 0.05).to(dev)
    s = (torch.rand(128, generator=g) * 3 + 0.5).to(dev)
    x = torch.randn(1, 40, 128, generator=g).to(dev)
    a8 = mq.QuantConfig(bitwidth=8)

    def make(weight):
        lin = torch.nn.Linear(128, 96, bias=False).to(dev)
        with torch.no_grad():
            lin.weight.copy_(weight)
        q = mq.QLinear.from_float(lin, a8, a8, a8).requires_grad_(False)
        q.set_scale_offset({"input": [-3.0, 3.0], "output": [-2.0, 2.0]}, "buffer")
        return q
    late, folded = make(w), make(w * s)
    with torch.no_grad():
        late(x)                                                     # first forward: grid from W
        stale = float(late.weight_quantizer.scale)
        late.set_input_channel_scale(s)
        got = late(x)
        want = folded(x / s)
    assert float(late.weight_quantizer.scale) == float(folded.weight_quantizer.scale) != stale
    assert torch.equal(got, want)
    # an explicitly set weight grid survives
    kept = make(w)
    kept.weight_quantizer.set_scale_offset_from_minmax(-0.3, 0.3, "buffer", dev)
    kept.set_input_channel_scale(s)
    assert kept.weight_quantizer._has_grid() and abs(float(kept.weight_quantizer.scale) - 0.6 / 255) < 1e-7


def test_prefill_attention_with_a_score_grid_too_wide_for_the_fixed_reference(dev):
    """with a 16-bit qk_bmm output grid whose span exceeds 2^96 in exp2 units the prefill kernel cannot use the grid's top
    as softmax reference and takes its running-maximum sweep (QK_OUT, not fixed_ref: mq_attention.hip) -- covered here against the
    numpy restatement of the reference."""
    from test_gpu_round2 import _attention_case, _grid_of
    from mobilequant_amd import ops
    S, heads, kv_heads = 128, 4, 2
    q, k, v, cos, sin, qk, pv = _attention_case(S, heads, kv_heads, seed=11)
    wide = O.QuantizerOracle(bitwidth=16)
    wide.set_from_minmax(F32(-700.0), F32(700.0))                   # (65535 steps * s / 8) * log2(e) = 252 > 96
    qk = (qk[0], qk[1], wide)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)                       # noqa: E731
    got = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids).cpu().numpy()
    step = float(pv[2].scale)
    diff = np.abs(got - want)
    assert np.isfinite(got).all() and diff.max() <= 1.001 * step and (diff > 0.5 * step).mean() < 0.02, (diff.max(), step)


def test_generate_reproduces_the_reference_models_free_running_greedy_stream_and_sampling_rule(dev):
    """DecodeEngine.generate is SimModel.generate's loop (sim_model.py:160-221).  Greedy: the 12-token continuation the reference's
    W8A8-simulated HFForCausalLM produces for an 8-token context when run free (tests/golden/generate_case.npz: every step's
    top-1 / top-2 margin >= 3 % of the logit span), with the context encoded by the prefill forward and token by token.  do_sample:
    next = multinomial(softmax(logits / temperature)) (:198-199) with a seeded device generator == the same rule applied by hand to
    the engine's own logits, step by step; EOS ends the stream AFTER the token is appended (:202-204)."""
    from conftest import load_npz
    from test_gpu_round2 import _decode_model
    from mobilequant_amd.decode import DecodeEngine
    m, _ = _decode_model(dev)
    z = load_npz("generate_case.npz")
    ctx, want = z["context"].tolist(), z["tokens"].tolist()
    eng = DecodeEngine(m, cache_len=64)
    assert eng.generate(ctx, len(want) - len(ctx)) == want
    assert eng.generate(ctx, len(want) - len(ctx), prefill=False) == want
    eng.capture()
    assert eng.generate(ctx, len(want) - len(ctx)) == want
    # EOS: the stream stops right behind the first occurrence of the EOS id, which is kept
    eos = want[len(ctx) + 3]
    cut = eng.generate(ctx, len(want) - len(ctx), eos_token_id=eos)
    first = len(ctx) + [t for t in want[len(ctx):]].index(eos)
    assert cut == want[:first + 1]
    # sampling: the reference's rule on the engine's own logits, identically seeded generators
    g1 = torch.Generator(device=dev).manual_seed(123)
    got = eng.generate(ctx, 10, do_sample=True, temperature=0.7, generator=g1)
    g2 = torch.Generator(device=dev).manual_seed(123)
    eng.reset()
    eng.prefill(ctx)
    by_hand = list(ctx)
    for _ in range(10):
        nxt = int(torch.multinomial(torch.softmax(eng.logits / 0.7, dim=-1), num_samples=1, generator=g2))
        by_hand.append(nxt)
        eng.step(nxt)
    assert got == by_hand and len(set(got[len(ctx):])) > 1


@pytest.mark.parametrize("S,heads,kv_heads", [(192, 4, 2), (256, 4, 4)])
def test_attention_kernels_against_the_exact_integer_oracle(dev, S, heads, kv_heads):
    """Where do index flips come from?  oracle.attention_sim(exact_int=True) carries both contractions out exactly over the indices --
    the integer kernels' arithmetic without their fast forms -- so:
      * the DECODE kernel (exact-divide quantizers, expf, true maximum) must agree with it on (practically) every output index: what
        is left is the exponential's last bits;
      * the PREFILL kernel's distance to it is what its reciprocal-multiply quantizers and exp2 forms cost: bounded
        here at twice the observed rate, far below the distance either has to the fp32-matmul form of the reference."""
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    D = 64
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, D, D, seed=S)
    exact = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv, exact_int=True)
    ref = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    step = float(pv[2].scale)
    st = _Stepper(dev, heads, kv_heads, D, D, S, cos, sin, qk, pv, pv[2], 1)
    dec = np.stack([st.step(t, q[t], k[t], v[t])[0] for t in range(S)])
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)                       # noqa: E731
    pre = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids).cpu().numpy()
    flips = {name: float((np.abs(got - exact) > 0.5 * step).mean()) for name, got in (("decode", dec), ("prefill", pre), ("fp32 matmul", ref))}
    worst = {name: float(np.abs(got - exact).max() / step) for name, got in (("decode", dec), ("prefill", pre))}
    print("index flips vs the exact-integer oracle:", flips, "max steps:", worst)
    assert worst["decode"] <= 1.001 and flips["decode"] <= 5e-5, (flips, worst)       # observed: 0 of 49 152 and 0 of 65 536
    assert worst["prefill"] <= 1.001 and flips["prefill"] <= 5e-4, (flips, worst)     # observed: 0 as well (fp32-matmul form: 1.5e-5)


def test_act_shifts_running_average_is_the_references(dev, tmp_path):
    """smoothquant.get_act_shifts (generate_act_scale_shift.py:97-149): the per-channel mid-range (max + min) / 2 of every hooked
    tensor and its 0.99 / 0.01 running average in SAMPLE ORDER, on the device (one HIP column reduction per tensor + three fp32
    elementwise ops) -- bit-identical to what the reference's hooks computed on the recorded stream of tests/golden/calib_stream.npz
    (6 ragged samples), through the statistic class and end to end through forward hooks on the device model; act_shifts.pth."""
    from conftest import load_npz
    from test_calibration_dist import _stream
    from toy_models import CalibToy
    from mobilequant_amd import smoothquant as S
    z = load_npz("calib_stream.npz")
    want = {k.split("|")[1]: z[k] for k in z.files if k.startswith("shift|")}
    assert len(want) == 6
    col = S.ActShiftCollector()
    for sample in _stream(z, "stream_pt"):
        for name, field, t in sample:
            if name in ("fc1", "fc2", "ln") and field in ("input", "output"):
                col.update(name, field, torch.from_numpy(t).to(dev))
    got = col.result()
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == torch.float32 and got[k].device.type == "cpu" and np.array_equal(got[k].numpy(), want[k]), k
    # end to end: hooks on the device copy of the toy stack, the reference's own samples (GPU forward: fp32 round-off apart)
    import json
    toy = CalibToy()
    toy.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("toy|")})
    toy = toy.to(dev).eval()
    samples = [torch.tensor([[int(t) for t in line.split()]]) for line in json.loads(str(z["ids_pt"]))]
    e2e = S.get_act_shifts(toy, samples)
    for k in want:
        assert np.allclose(e2e[k].numpy(), want[k], rtol=1e-5, atol=1e-6), k
    S.save_act_shifts(str(tmp_path / "act_shifts.pth"), e2e)
    back = torch.load(str(tmp_path / "act_shifts.pth"), map_location="cpu")
    assert list(back) == list(e2e) and all(torch.equal(back[k], e2e[k]) for k in e2e)


def test_one_e2equant_training_step_loss_and_every_gradient_vs_the_reference(dev):
    """f3 closed: ONE inner step of e2equant (algorithm.py:727-745) under the deployment recipe's flags (--lwc --let --lrl
    --deactive_amp, fp32, 4-bit per-channel weights) on a decoder layer, with this package's modules: HIP fake-quant forward AND
    backward (STE, clamp mask, LSQ gradients to scale / offset), LWC through the HIP range reduction, LET through
    smooth_lm_temporary -- against the loss and the gradient of every trainable tensor (18 LWC bound factors, 5 LET scales, 20 + 20
    quantizer scales / offsets) the reference's autograd produced for the same layer, input and ranges (tests/golden/train_step.npz).
    The GPU's fp32 matmuls sum in another order than the CPU's, so a value that sits on a rounding boundary can take the other
    grid point and move its STE mask: gradients are held to direction (cosine) and size, not bits."""
    import json
    import mobilequant_amd as mq
    from conftest import load_npz
    from mobilequant_amd.llama import DecoderLayer, LlamaShape, rope_tables
    from toy_models import apply_mixed_precision
    z = load_npz("train_step.npz")
    shape = LlamaShape(hidden=64, layers=1, heads=4, kv_heads=4, head_dim=16, ffn=96, vocab=50, eps=1e-5, max_pos=64)
    layer = DecoderLayer(shape)
    for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj, layer.mlp.w1, layer.mlp.w2,
                layer.mlp.w3):
        lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features))
    layer.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd|")})
    layer = layer.to(dev)
    mq.create_sim_qmodel(layer, mq.QuantConfig(bitwidth=4, is_per_channel=True), mq.QuantConfig(bitwidth=8))
    apply_mixed_precision(layer, mq)
    for name, mod in layer.named_modules():
        if isinstance(mod, mq.QRMSNorm):
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
    mq.set_scale_and_offset(layer, json.loads(str(z["act"])), "parameter")
    for name, mod in layer.named_modules():                          # enable_quant with --lwc (algorithm.py:325-351)
        if isinstance(mod, (mq.QLinear, mq.QRMSNorm)):
            mod.weight_quantizer.enable_lwc(mod.weight)
    for k in z.files:
        if k.startswith("let|"):
            layer.register_parameter(k[4:], torch.nn.Parameter(T_(z[k], dev)))
    for name in ("qkv", "fc1", "out", "fc2"):
        layer.register_parameter(f"{name}_smooth_shift", torch.nn.Parameter(torch.zeros(96 if name == "fc2" else 64, device=dev)))
    for p in layer.parameters():
        p.requires_grad_(True)
    cfg = type("Cfg", (), dict(shared_attention_norm=False, num_linears_per_mlp=3))()
    x, y_fp = T_(z["x"], dev), T_(z["y_fp"], dev)
    S = x.shape[1]
    cos, sin = (t[:S].to(dev) for t in rope_tables(shape))
    mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
    mq.smooth_lm_temporary(layer, cfg, True, False)
    y_q = layer(x, cos, sin, mask)
    loss = torch.nn.MSELoss()(y_fp, y_q)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) <= 2e-3 * float(z["loss"]), (float(loss), float(z["loss"]))
    d = (y_q.detach().cpu().numpy() - z["y_q"])
    assert np.abs(d).max() <= 0.05 * np.ptp(z["y_q"]) and np.median(np.abs(d)) <= 1e-4 * np.ptp(z["y_q"])
    named = dict(layer.named_parameters())
    rows = []
    for k in z.files:
        if not k.startswith("grad|"):
            continue
        name, want = k[5:], z[k].astype(np.float64).ravel()
        assert name in named and named[name].grad is not None, name
        got = named[name].grad.detach().cpu().numpy().astype(np.float64).ravel()
        assert got.shape == want.shape and np.isfinite(got).all(), name
        group = next(t for t in ("bound_factor", "smooth_scale", "quantizer.scale", "quantizer.offset") if t in name)
        rows.append((group, name, np.linalg.norm(want), np.linalg.norm(got), np.linalg.norm(got - want)))
    assert len(rows) == 63
    # observed (round 3): LET scales and LWC bound factors agree to 6e-6 relative; quantizer scales / offsets to 3e-2 at worst (sums
    # with heavy cancellation over 16-bit grids), except gradients that ARE cancellation residue -- seven orders below their group's
    # largest (pv_bmm.input2: 1.5e-5 against 1.7e+2) -- which are only required to stay at that level
    for group, bar in (("smooth_scale", 1e-4), ("bound_factor", 1e-4), ("quantizer.scale", 0.06), ("quantizer.offset", 0.06)):
        top = max(r[2] for r in rows if r[0] == group)
        for _, name, nw, ng, ne in rows:
            if _ != group:
                continue
            if nw <= 1e-6 * top:
                assert ng <= 1e-5 * top, (name, nw, ng)
            else:
                assert ne <= bar * nw, (name, nw, ng, ne / nw)


def T_(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_torch_ops_are_the_c_abi_kernels(dev):
    """torch.ops.mobilequant_amd.* (torch.library registration) dispatch to the same kernels as the ctypes path: identical
    tensors for fake_quant / quantize / minmax / w8a8_linear / w4a8_linear on random inputs; minmax_update_ mutates in place."""
    import mobilequant_amd.torch_ops  # noqa: F401
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_I8, MQ_U8
    ns = torch.ops.mobilequant_amd
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(24, 256, generator=g) * 2).to(dev)
    s, o = torch.tensor([0.03], device=dev), torch.tensor([117.0], device=dev)
    assert torch.equal(ns.fake_quant(x, s, o, 0.0, 255.0), ops.fake_quant(x, s, o, 0.0, 255.0))
    q, rs = ns.quantize(x, s, o, 0.0, 255.0, 128)
    q2, rs2 = ops.quantize(x, s, o, 0.0, 255.0, q_dtype=MQ_I8, shift=128, rows=24, want_row_sum=True)
    assert torch.equal(q, q2) and torch.equal(rs, rs2)
    mm = ns.minmax(x, False)
    assert float(mm[0]) == float(x.min()) and float(mm[1]) == float(x.max())
    mc = ns.minmax(x, True)
    assert torch.equal(mc[0], x.min(0).values) and torch.equal(mc[1], x.max(0).values)
    running = torch.stack((torch.full((256,), float("inf"), device=dev), torch.full((256,), float("-inf"), device=dev)))
    ns.minmax_update_(running, x)
    assert torch.equal(running, mc)
    w = torch.randint(-128, 128, (64, 256), generator=g, dtype=torch.int8).to(dev)
    colsum = w.to(torch.int32).sum(1).to(torch.int32)
    ws, wo = torch.full((1,), 0.01, device=dev), torch.full((1,), 3.0, device=dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(s, o, 128, ws, wo, 128, colsum, 256)
    assert torch.equal(ns.w8a8_linear(q, rs, w, alpha, wzp, ct), ops.int8_linear(q, w, rs, alpha, wzp, ct))
    nib = torch.randint(0, 16, (64, 256), generator=g, dtype=torch.uint8).to(dev)
    packed = ns.pack_w4(nib)
    assert torch.equal(packed, ops.pack_w4(nib))
    a4, z4, c4 = ops.linear_epilogue_prepare(s, o, 128, ws, torch.full((1,), 5.0, device=dev), 0, nib.to(torch.int32).sum(1).to(torch.int32), 256)
    assert torch.equal(ns.w4a8_linear(q, rs, packed, a4, z4, c4), ops.int8_linear(q, packed, rs, a4, z4, c4, w4=True))


@pytest.mark.parametrize("rot", [16, 32])
def test_prefill_attention_with_partial_rotary_vs_oracle(dev, rot):
    """mq_attention_quant with RoPE on the first rot_dim of 64 head dims (StableLM-2: partial_rotary_factor 0.25, hf_model.py:489-500)
    against the numpy restatement; the rest of the kernel is the full-rotary one."""
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    S, heads, kv_heads = 128, 4, 4
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, 64, rot, seed=rot)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)                       # noqa: E731
    got = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids).cpu().numpy()
    step = float(pv[2].scale)
    diff = np.abs(got - want)
    assert np.isfinite(got).all() and diff.max() <= 1.001 * step and (diff > 0.5 * step).mean() < 0.005, (diff.max(), step)


# ---- the 128-column generated kernels on fragment-blocked activations -------------------------------------------------------------
def _to_tiled(a_q):
    """row-major int8 [M, K] -> the fragment-blocked image of mq_quantize_tiled ([ceil16(M), K]; include/mobilequant_amd.h)."""
    M, K = a_q.shape
    Mp = (M + 15) // 16 * 16
    pad = torch.zeros((Mp, K), dtype=torch.int8, device=a_q.device)
    pad[:M] = a_q
    return pad.view(Mp // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(Mp, K)


def _gemm_operands(dev, M, N, K, seed, w_zp_zero=False, bias=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a_q = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(dev)
    w_q = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g).to(dev)
    a_rs = a_q.to(torch.int32).sum(dim=1, dtype=torch.int32)
    alpha = (torch.rand(N, generator=g) * 2e-4 + 1e-5).to(dev)
    w_zp = (torch.zeros(N, dtype=torch.int32) if w_zp_zero else torch.randint(-100, 100, (N,), dtype=torch.int32, generator=g)).to(dev)
    col_term = torch.randint(-50000, 50000, (N,), dtype=torch.int32, generator=g).to(dev)
    b = (torch.randn(N, generator=g) * 0.3).to(dev) if bias else None
    return a_q, w_q, a_rs, alpha, w_zp, col_term, b


@pytest.mark.parametrize("M,N,K,tile", [(2048, 2048, 2048, 0), (2048, 2048, 5632, 128), (2048, 2048, 2048, 256), (300, 256, 768, 128),
                                        (300, 256, 768, 256), (129, 384, 1024, 0), (4096, 2048, 1024, 0),
                                        # 512 = 256-row tiles with the K loop split over two workgroups (round 4; served when both halves
                                        # of every tile are resident at once, otherwise the call falls back to the unsplit tile)
                                        (2048, 2048, 5632, 512), (2048, 2048, 2048, 512), (1900, 2048, 1536, 512), (300, 1024, 2048, 512),
                                        (2048, 2048, 16384, 512), (4096, 2048, 2048, 512)])
def test_tiled_residual_gemm_is_the_rowmajor_residual_gemm_bit_for_bit(dev, M, N, K, tile):
    """mq_w8a8_linear_tiled_residual (generated ISA, 128 x 128 / 256 x 128 tiles, fragment-blocked activations) against
    mq_w8a8_linear_residual (the C++ kernel the golden decode / layer cases pin): x + Q16(linear), same bits."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    for variant, (zp0, bias) in enumerate(((False, True), (True, False))):
        a_q, w_q, a_rs, alpha, w_zp, col_term, b = _gemm_operands(dev, M, N, K, 11 * M + N + K + variant, zp0, bias)
        resid = torch.randn(M, N, device=dev)
        so = torch.tensor([3.1e-4], device=dev)
        oo = torch.tensor([32768.0 if variant == 0 else 30111.0], device=dev)
        kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0)
        want = ops.int8_linear(a_q, w_q, None if zp0 else a_rs, alpha, w_zp, col_term, b, resid=resid, **kw)
        if L.load().mq_gemm_set_residual_tile(tile) != 0:       # 512 (split-K, a measured negative): experiment builds only
            L.load().mq_gemm_set_residual_tile(0)
            pytest.skip("split-K residual tiles are compiled with `python -m mobilequant_amd.build --experiments` only")
        try:
            got = ops.int8_linear(_to_tiled(a_q), w_q, None if zp0 else a_rs, alpha, w_zp, col_term, b, resid=resid, a_tiled_rows=M, **kw)
        finally:
            L.load().mq_gemm_set_residual_tile(0)
        torch.cuda.synchronize()
        assert got.shape == want.shape and torch.equal(got, want), (variant, (got - want).abs().max().item())
        q = torch.round((want - resid) / so + oo)            # the grid is actually exercised (not saturated everywhere)
        assert q.min() < 20000 and q.max() > 45000


@pytest.mark.parametrize("M,K,ends", [(2048, 2048, (2048, 2304, 2560)), (300, 768, (128, 256)), (513, 1024, (384,)), (300, 768, (256, 512, 640)),
                                      (1999, 1024, (640,))])
def test_tiled_segmented_gemm_is_the_rowmajor_segmented_gemm_bit_for_bit(dev, M, K, ends):
    """mq_w8a8_linear_tiled_segmented (q | k | v on 256 x 128 tiles of generated ISA, one output grid per column segment) against
    mq_w8a8_linear_segmented."""
    from mobilequant_amd import ops
    N = ends[-1]
    a_q, w_q, a_rs, alpha, w_zp, col_term, b = _gemm_operands(dev, M, N, K, 5 * M + K)
    alpha = alpha * 0.02
    grids = [(torch.tensor([0.011 * (i + 1)], device=dev), torch.tensor([100.0 + 20 * i], device=dev)) for i in range(len(ends))]
    want = ops.int8_linear_segmented(a_q, w_q, a_rs, alpha, w_zp, col_term, b, ends, grids)
    got = ops.int8_linear_segmented(_to_tiled(a_q), w_q, a_rs, alpha, w_zp, col_term, b, ends, grids, a_tiled_rows=M)
    torch.cuda.synchronize()
    assert torch.equal(got, want), (got.int() - want.int()).abs().max().item()
    import mobilequant_amd._lib as L                # N % 160 == 0 with at most 256 tiles ran on 128 x 160 tiles: also the 256 x 128 kernel
    L.load().mq_gemm_set_segmented_tile(128)
    try:
        got128 = ops.int8_linear_segmented(_to_tiled(a_q), w_q, a_rs, alpha, w_zp, col_term, b, ends, grids, a_tiled_rows=M)
    finally:
        L.load().mq_gemm_set_segmented_tile(0)
    assert torch.equal(got128, want)
    assert 5 < want.float().mean() < 250 and want.min() == 0 and want.max() == 255


def test_gated_lookup_tiled_is_the_rowmajor_lookup_in_the_fragment_blocked_layout(dev):
    from mobilequant_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    table = torch.randint(-128, 128, (65536,), dtype=torch.int8, generator=g).to(dev)
    for rows, cols in ((2048, 5632), (37, 128), (130, 8192 + 64)):
        a = torch.randint(0, 256, (rows, cols), dtype=torch.uint8, generator=g).to(dev)
        b = torch.randint(0, 256, (rows, cols), dtype=torch.uint8, generator=g).to(dev)
        q, rs = ops.gated_lookup(a, b, table)
        qt, rst = ops.gated_lookup(a, b, table, tiled=True)
        torch.cuda.synchronize()
        assert torch.equal(rs, rst)
        Mp = (rows + 15) // 16 * 16
        back = qt.view(Mp // 16, cols // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]
        assert torch.equal(back, q)


def test_tiled_index_gemm_on_128_column_tiles_is_the_rowmajor_gemm(dev):
    """mq_w8a8_linear_tiled for an N that does not tile by 176 (Gemma's FFN, N = 16384 here 1024): u8 index output through the
    128-column generated kernel == mq_w8a8_linear on the row-major image."""
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    for M, N, K in ((2048, 1024, 2048), (200, 384, 768)):
        a_q, w_q, a_rs, alpha, w_zp, col_term, b = _gemm_operands(dev, M, N, K, M + N)
        so, oo = torch.tensor([0.9], device=dev), torch.tensor([131.0], device=dev)
        kw = dict(out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8)
        want = ops.int8_linear(a_q, w_q, a_rs, alpha * 20, w_zp, col_term, b, **kw)
        got = ops.int8_linear(_to_tiled(a_q), w_q, a_rs, alpha * 20, w_zp, col_term, b, a_tiled_rows=M, **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and want.min() == 0 and want.max() == 255


# ---- prefill attention at head_dim 256 (Gemma: 8 heads, 1 KV head) -------------------------------------------------------------------
@pytest.mark.parametrize("D,S,heads,kv_heads,qk_out_bits,pv_out_bits", [(256, 64, 2, 1, 16, 8), (256, 200, 8, 1, 16, 8), (256, 320, 4, 2, 16, 8),
                                                                        (256, 128, 2, 2, 0, 0), (128, 200, 4, 2, 16, 8), (128, 128, 2, 1, 0, 0)])
def test_prefill_attention_head_dim_256_vs_oracle(dev, D, S, heads, kv_heads, qk_out_bits, pv_out_bits):
    """mq_attention_quant at head_dim 256 (four MFMA k-steps per score tile, 16 output d-tiles, sum_t v from the prep kernel's prefix
    sums) against the numpy restatement of hf_model.py:486-534 and against its exact-integer form; the int8 output image in both
    layouts is the fp32 output's index image."""
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, D, D, seed=7 * S + heads, qk_out_bits=qk_out_bits, pv_out_bits=pv_out_bits)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)                       # noqa: E731
    got_t = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, head_dim=D)
    got = got_t.cpu().numpy()
    assert got.shape == want.shape and np.isfinite(got).all()
    diff = np.abs(got - want)
    span = float(want.max() - want.min())
    if pv_out_bits == 8:
        step = float(pv[2].scale)
        assert diff.max() <= 1.001 * step, (diff.max(), step)
        assert (diff > 0.5 * step).mean() < 0.02
        exact = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv, exact_int=True)
        assert (np.abs(got - exact) > 0.5 * step).mean() <= 5e-4
        # the int8 images (row-major / fragment-blocked) carry exactly the indices of the fp32 output, the row sums their sums
        idx = torch.round(got_t / float(pv[2].scale) + float(pv[2].offset)).to(torch.int32) - 128
        for tiled in (False, True):
            Mp = (S + 15) // 16 * 16
            img = torch.zeros((Mp if tiled else S, heads * D), dtype=torch.int8, device=dev)
            rs = torch.zeros(S, dtype=torch.int32, device=dev)
            ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, image=(img, rs, 0, 128, tiled), want_out=False, head_dim=D)
            torch.cuda.synchronize()
            back = img.view(Mp // 16, heads * D // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, heads * D)[:S] if tiled else img
            assert torch.equal(back.to(torch.int32), idx), tiled
            assert torch.equal(rs, idx.sum(dim=1, dtype=torch.int32))
    else:
        assert diff.max() <= 2e-3 * span, (diff.max(), span)
    assert np.median(diff) <= 2e-4 * span


# ---- chunked prefill: the fused attention continuing a cache of K / vT images ----------------------------------------------------------
@pytest.mark.parametrize("D,heads,kv_heads,chunks", [(64, 4, 2, (128, 64, 100)), (256, 2, 1, (64, 128, 37)), (64, 2, 2, (192, 192)),
                                                     (128, 4, 1, (64, 64, 70))])
def test_attention_quant_cache_continuation_is_the_single_shot_result(dev, D, heads, kv_heads, chunks):
    """mq_attention_quant with pos0 > 0: a sequence fed in chunks (each but the last a multiple of 64) through caller-owned K / vT image
    caches gives, row for row, the bits of the single call over the whole sequence (the same key blocks in the same order), and both
    stay on the oracle."""
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    S = sum(chunks)
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, D, D, seed=S + D)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                       # noqa: E731
    whole = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids, head_dim=D)
    cache = ops.attention_image_cache(kv_heads, D, S + 7, dev)
    outs, p0 = [], 0
    for n in chunks:
        sl = slice(p0, p0 + n)
        outs.append(ops.attention_quant(t(q[sl]), t(k[sl]), t(v[sl]), t(cos[sl]), t(sin[sl]), heads, kv_heads, grids, head_dim=D, cache=cache, pos0=p0))
        p0 += n
    got = torch.cat(outs)
    torch.cuda.synchronize()
    assert torch.equal(got, whole), float((got - whole).abs().max())
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    assert np.abs(got.cpu().numpy() - want).max() <= 1.001 * float(pv[2].scale)
    with pytest.raises(RuntimeError):
        ops.attention_quant(t(q[:64]), t(k[:64]), t(v[:64]), t(cos[:64]), t(sin[:64]), heads, kv_heads, grids, head_dim=D, cache=cache, pos0=32)


def test_chunked_prefill_through_the_fused_model_matches_the_single_forward(dev):
    """LlamaForCausalLM.new_image_cache + fuse_decoder_layer: a 192-token context fed as 128 + 64 reproduces the single fused forward's
    logits bit for bit (every per-row computation is the same), on the 2-layer W8A8 model of decode_case.npz."""
    from test_gpu_round2 import _decode_model
    from mobilequant_amd import llama
    import dataclasses
    m, z = _decode_model(dev)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=256))      # the golden model's table stops at 64 positions
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, m.shape.vocab, (1, 192), generator=g).to(dev)
    S = ids.shape[1]
    with torch.no_grad():
        assert llama.fuse_decoder_layer(m) == 2
        whole = m(ids)
        cache = m.new_image_cache(1, S)
        parts = [m(ids[:, :64], cache=cache, pos=0), m(ids[:, 64:S], cache=cache, pos=64)]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, dim=1), whole)


@pytest.mark.parametrize("rows,cols,layernorm", [(2048, 2048, False), (100, 2048, False), (2048, 1024, True), (77, 4096, True), (64, 2048, False)])
def test_image_only_tiled_norm_is_the_rowmajor_image_in_the_fragment_blocked_layout(dev, rows, cols, layernorm):
    """norm_tiled8_kernel (image-only, fragment-blocked output: eight rows per workgroup, LDS-staged whole-line stores) against the
    generic norm kernel: same int8 indices (the row-major image, permuted) and row sums, RMSNorm and LayerNorm, ragged row counts."""
    from mobilequant_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 1.7).to(dev)
    w = (1.0 + 0.1 * torch.randn(cols, generator=g)).to(dev)
    b = (0.1 * torch.randn(cols, generator=g)).to(dev) if layernorm else None
    gi = (torch.tensor([10.0 / 65535], device=dev), torch.tensor([32768.0], device=dev), 0.0, 65535.0)
    go = (torch.tensor([8.0 / 255], device=dev), torch.tensor([128.0], device=dev), 0.0, 255.0)
    _, q, rs, shift, _ = ops.rmsnorm_quant(x, w, b, 1e-5, gi, go, emit_int8=True, layernorm=layernorm, emit_tiled=False, want_y=False, emit_rowmajor=True)
    import mobilequant_amd._lib as L
    for knob in (0, 4, 8):                     # rows per workgroup: by shape (default) / four (two 512-thread workgroups per CU) / eight
        L.load().mq_norm_tiled_set_rows(knob)
        try:
            _, _, rst, shift_t, qt = ops.rmsnorm_quant(x, w, b, 1e-5, gi, go, emit_int8=True, layernorm=layernorm, emit_tiled=True, want_y=False, emit_rowmajor=False)
            torch.cuda.synchronize()
        finally:
            L.load().mq_norm_tiled_set_rows(0)
        assert shift == shift_t and torch.equal(rs, rst)
        Mp = (rows + 15) // 16 * 16
        back = qt.view(Mp // 16, cols // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]
        assert torch.equal(back, q.view(rows, cols))


@pytest.mark.parametrize("rows,cols", [(2048, 2048), (100, 2048), (333, 1024), (64, 4096), (2048, 3072)])
def test_staged_quantize_tiled_writes_the_same_image(dev, rows, cols):
    """mq_quantize_tiled's LDS-staged eight-row kernel (fp32, 1024 <= cols <= 4096) against the lane-per-fragment kernel it replaces there:
    the same bytes (padding rows included where both write them) and row sums."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) * 2.0).to(dev)
    sc, of = torch.tensor([0.031], device=dev), torch.tensor([131.0], device=dev)
    L.load().mq_quantize_tiled_set_staged(0)
    try:
        q0, rs0 = ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128)
    finally:
        L.load().mq_quantize_tiled_set_staged(1)
    Mp = q0.shape[0]
    un = lambda q: q.view(Mp // 16, cols // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]      # noqa: E731
    for knob in (0, 4, 8):                     # rows per workgroup of the staged kernel: by shape (default) / four / eight
        L.load().mq_quantize_tiled_set_rows(knob)
        try:
            q1, rs1 = ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128)
            torch.cuda.synchronize()
        finally:
            L.load().mq_quantize_tiled_set_rows(0)
        assert torch.equal(rs0, rs1)
        assert torch.equal(un(q0), un(q1))


@pytest.mark.parametrize("signed", [False, True])
def test_packed_convert_image_kernels_on_saturating_and_non_finite_inputs(dev, signed):
    """The staged norm / quantize kernels form their bytes with v_med3 + v_cvt_pk_u8_f32 + v_sad_u8 (mq_common.h image_u8f / image_pack4):
    same bytes and row sums as the generic kernels on values far outside the grid, on a signed grid (shift 0), and with NaN / +-inf
    elements -- a non-finite element saturates to qmin in mq_quantize_tiled and poisons its whole row in the norm (every index qmin)."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    rows, cols = 200, 2048
    g = torch.Generator(device="cpu").manual_seed(17 + signed)
    x = torch.randn(rows, cols, generator=g) * 3.0
    x[3, 5], x[3, 900], x[77, 0], x[150, 2047] = float("nan"), 1e30, float("inf"), float("-inf")
    x[10] *= 1e6
    x = x.to(dev)
    un = lambda q, Mp: q.view(Mp // 16, cols // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]   # noqa: E731
    qmin, qmax, shift, off = (-128.0, 127.0, 0, 3.0) if signed else (0.0, 255.0, 128, 131.0)
    sc, of = torch.tensor([0.031], device=dev), torch.tensor([off], device=dev)
    L.load().mq_quantize_tiled_set_staged(0)
    try:
        q0, rs0 = ops.quantize_tiled(x, sc, of, qmin, qmax, shift)
    finally:
        L.load().mq_quantize_tiled_set_staged(1)
    q1, rs1 = ops.quantize_tiled(x, sc, of, qmin, qmax, shift)
    torch.cuda.synchronize()
    assert torch.equal(rs0, rs1) and torch.equal(un(q0, q0.shape[0]), un(q1, q1.shape[0]))
    assert int(un(q1, q1.shape[0])[3, 5]) == int(qmin) - shift and int(un(q1, q1.shape[0])[3, 900]) == int(qmax) - shift
    w = (1.0 + 0.1 * torch.randn(cols, generator=g)).to(dev)
    gi = (torch.tensor([40.0 / 65535], device=dev), torch.tensor([32768.0], device=dev), 0.0, 65535.0)
    go = (torch.tensor([8.0 / 255], device=dev), torch.tensor([off], device=dev), qmin, qmax)
    for ln in (False, True):
        b = (0.1 * torch.randn(cols, generator=g)).to(dev) if ln else None
        for gin in (gi, None):
            _, q, rs, sh, _ = ops.rmsnorm_quant(x, w, b, 1e-5, gin, go, emit_int8=True, layernorm=ln, emit_tiled=False, want_y=False, emit_rowmajor=True)
            _, _, rst, sht, qt = ops.rmsnorm_quant(x, w, b, 1e-5, gin, go, emit_int8=True, layernorm=ln, emit_tiled=True, want_y=False, emit_rowmajor=False)
            torch.cuda.synchronize()
            assert sh == sht and torch.equal(rs, rst)
            back = un(qt, (rows + 15) // 16 * 16)
            assert torch.equal(back, q.view(rows, cols))
            for r in ((3, 77, 150) if gin is not None else (3,)):   # no input quantizer: an inf row has 1 / rms = 0, only inf * 0 is NaN
                assert bool((back[r] == int(qmin) - sh).all())


def test_decode_engine_switches_to_the_split_attention_graph_on_a_long_cache(dev):
    """DecodeEngine(attn_splits=None) records two step graphs when the cache is longer than LONG_FROM and replays the split-attention one
    from LONG_FROM cached positions on: the logits are those of a one-workgroup-per-head engine, bit for bit, on both sides of the
    switch (the split launch's int64 partials are combined in ticket order)."""
    import dataclasses
    from test_gpu_round2 import _decode_model
    from mobilequant_amd import llama
    from mobilequant_amd.decode import DecodeEngine
    m, _ = _decode_model(dev)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=1024))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    auto = DecodeEngine(m, cache_len=1024, launches=5)
    one = DecodeEngine(m, cache_len=1024, attn_splits=1, launches=5)
    assert auto.auto_splits and not one.auto_splits
    for eng in (auto, one):
        eng.fill_cache_random(DecodeEngine.LONG_FROM - 2, seed=3)
        eng.capture()
    assert auto.graph_long is not None and one.graph_long is None
    for t in (5, 17, 40, 3, 90):                         # two steps below LONG_FROM, three from it on
        a = auto.step(t).clone()
        b = one.step(t).clone()
        torch.cuda.synchronize()
        assert torch.equal(a, b), (auto._host_pos, float((a - b).abs().max()))
    assert auto._host_pos == DecodeEngine.LONG_FROM + 3


@pytest.mark.parametrize("S,heads,kv_heads", [(704, 4, 1), (130, 2, 2), (1024, 2, 1)])
def test_attention_exponential_cache_depth_does_not_change_a_bit(dev, S, heads, kv_heads):
    """head_dim 64: the sweep-1 exponentials of the last key blocks are reused by sweep 2 (deep cache: four blocks in the LDS + five in
    registers; small cache: two in the LDS).  Both configurations -- and so every mix of recomputed / parked blocks, S = 704 has eleven
    key blocks -- give the same image and row sums, and the deep one stays on the oracle."""
    import mobilequant_amd._lib as L
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    q, k, v, cos, sin, qk, pv = _case(S, heads, kv_heads, 64, 64, seed=S)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)                       # noqa: E731
    outs = []
    for mode in (1, 0):
        L.load().mq_attention_set_cache(mode)
        try:
            outs.append(ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids))
        finally:
            L.load().mq_attention_set_cache(0)
    # ... and the q rows prepared inside the attention workgroups (default) against the prep kernel's q image
    L.load().mq_attention_set_fused_q(0)
    try:
        outs.append(ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv_heads, grids))
    finally:
        L.load().mq_attention_set_fused_q(1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    want = O.attention_sim(q, k, v, cos, sin, heads, kv_heads, qk, pv)
    assert np.abs(outs[1].cpu().numpy() - want).max() <= 1.001 * float(pv[2].scale)


@pytest.mark.parametrize("S,heads,kv_heads", [(200, 4, 2), (1024, 8, 1)])
def test_attention_q_rows_prepared_in_the_core_kernel_from_gemm_indices(dev, S, heads, kv_heads):
    """Index input (the fused q|k|v GEMM's uint8 output indices, ragged S): the attention workgroups dequantise, rotate and quantise their
    own q rows (mq_attention_set_fused_q(1), default) -- same fp32 output, int8 image and row sums as with the prep kernel's q image."""
    import mobilequant_amd._lib as L
    from test_gpu_round2 import _grid_of
    from mobilequant_amd import ops
    _, _, _, cos, sin, qk, pv = _case(S, heads, kv_heads, 64, 64, seed=S + 1)
    grids = dict(qk_a=_grid_of(qk[0], dev), qk_b=_grid_of(qk[1], dev), qk_out=_grid_of(qk[2], dev), pv_a=_grid_of(pv[0], dev),
                 pv_b=_grid_of(pv[1], dev), pv_out=_grid_of(pv[2], dev))
    g = torch.Generator(device="cpu").manual_seed(S)
    idx = torch.randint(0, 256, (S, (heads + 2 * kv_heads) * 64), dtype=torch.uint8, generator=g).to(dev)
    in_grids = tuple((torch.tensor([0.03 + 0.01 * i], device=dev), torch.tensor([127.0 + i], device=dev)) for i in range(3))
    res = []
    for on in (0, 1):
        L.load().mq_attention_set_fused_q(on)
        try:
            img = torch.zeros(S, heads * 64, dtype=torch.int8, device=dev)
            rs = torch.zeros(S, dtype=torch.int32, device=dev)
            out = ops.attention_quant(None, None, None, torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev), heads, kv_heads, grids,
                                      image=(img, rs, 0, 128, False), qkv_idx=(idx, in_grids))
            res.append((out, img, rs))
        finally:
            L.load().mq_attention_set_fused_q(1)
    torch.cuda.synchronize()
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    assert float(res[1][0].abs().max()) > 0


@pytest.mark.parametrize("M,N,K,zp0", [(2048, 5632, 2048, False), (1700, 5632, 2048, True), (1537, 5632, 2048, False), (2048, 3072, 1024, False),
                                       (1900, 4096, 768, True)])
def test_gated_pair_with_the_lookup_in_the_epilogue_is_pair_plus_lookup(dev, M, N, K, zp0):
    """mq_w8a8_linear_tiled_gated (w1 launch + w3 launch whose generated epilogue does the table lookup) against
    mq_w8a8_linear_tiled_pair + mq_gated_lookup_tiled: the same fragment-blocked image of w2's input and the same row sums."""
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    g = torch.Generator(device="cpu").manual_seed(M)
    halves = []
    a_q = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(dev)
    a_rs = a_q.to(torch.int32).sum(dim=1, dtype=torch.int32)
    for h in range(2):
        _, w_q, _, alpha, w_zp, col_term, b = _gemm_operands(dev, 16, N, K, 50 + h + M, zp0, bias=(h == 0))
        halves.append(dict(w=w_q, alpha=alpha * 0.02, w_zp=w_zp, col_term=col_term, bias=b,
                           out_scale=torch.tensor([0.011 * (h + 1)], device=dev), out_offset=torch.tensor([120.0 + 9 * h], device=dev)))
    table = torch.randint(-128, 128, (65536,), dtype=torch.int8, generator=g).to(dev)
    a_t = _to_tiled(a_q)
    rs_in = None if zp0 else a_rs
    if N % 176 == 0:
        ia, ib = ops.int8_linear_pair(a_t, M, rs_in, halves[0], halves[1])
    else:                                       # N = 3072 / 4096: the 256 x 128 tile variants (w1 on fr128, w3 on frg128)
        ia, ib = (ops.int8_linear(a_t, h["w"], rs_in, h["alpha"], h["w_zp"], h["col_term"], h["bias"], out_scale=h["out_scale"],
                                  out_offset=h["out_offset"], out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8, a_tiled_rows=M) for h in halves)
    q_want, rs_want = ops.gated_lookup(ia, ib, table, tiled=True)
    q_got, rs_got = ops.int8_linear_gated(a_t, M, rs_in, halves[0], halves[1], table)
    torch.cuda.synchronize()
    assert 5 < ia.float().mean() < 250 and 5 < ib.float().mean() < 250
    assert torch.equal(rs_got, rs_want)
    Mp = q_want.shape[0]
    un = lambda q: q.view(Mp // 16, N // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(Mp, N)[:M]      # noqa: E731
    assert torch.equal(un(q_got), un(q_want))
