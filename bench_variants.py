#!/usr/bin/env python3
"""The non-headline legs of the benchmark (bench.py used to run all of them in every driver run -- 173 s at 4 % GPU
busy): module forward, FFN pair GEMM, whole decoder layers at prefill (three families, W8 / W4), one e2equant training step, the 22-layer
model prefill, the calibration reductions, linears-only decode, BASELINE.json configs[2] / [3] GEMMs.

    python bench_variants.py [bench.py flags]      ==      python bench.py --variants [flags]

prints bench.py's JSON line with every leg inside `variants`.  bench.py itself imports bench_other_configs from here (configs[2] / [3]
belong to the headline line); the rest runs only on request.  Shared helpers (Step, event_time, calibration_run, ...) live in bench.py."""
from __future__ import annotations

import os
import sys
import time  # noqa: F401

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench as _b  # noqa: E402
from bench import (INT8_MFMA_PEAK_TOPS, K, M, N, OPS_PER_STEP, bench_decode_full, bench_minmax, calibration_run, event_time,  # noqa: E402,F401
                   _stub_gemms)


def bench_decode_linears(dev, w4=False):
    """TinyLlama-1.1B decode, linears only: per layer quantize(x) -> GEMV qkv (2048->2560) -> quantize -> GEMV o
    (2048->2048) -> quantize -> GEMV w1|w3 (2048->11264) -> quantize -> GEMV w2 (5632->2048), 22 layers with their own
    int8 weights (0.97 GB streamed per token), one hipGraph per token.  Attention, norms and sampling are outside the
    hot path of this repository and are not included."""
    import mobilequant_amd as mq
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_F32, MQ_I8
    g = torch.Generator(device="cpu").manual_seed(7)
    shapes = [(2048, 2560), (2048, 2048), (2048, 11264), (5632, 2048)]      # (K, N) per layer
    layers = []
    aq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); aq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)
    oq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); oq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)
    for _ in range(22):
        lw = []
        for Kk, Nn in shapes:
            if w4:      # packed unsigned nibbles, zero point 8
                nib = torch.randint(0, 16, (Nn, Kk), dtype=torch.uint8, generator=g).to(dev)
                colsum = nib.to(torch.int32).sum(1).to(torch.int32)
                w8 = ops.pack_w4(nib)
                wscale = torch.full((1,), 3e-3, device=dev); woff = torch.full((1,), 8.0, device=dev)
                alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, 128, wscale, woff, 0, colsum, Kk)
            else:
                w8 = torch.randint(-128, 128, (Nn, Kk), dtype=torch.int8, generator=g).to(dev)
                colsum = w8.to(torch.int32).sum(1).to(torch.int32)
                wscale = torch.full((1,), 2e-4, device=dev); woff = torch.full((1,), 128.0, device=dev)
                alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, 128, wscale, woff, 128, colsum, Kk)
            lw.append((w8, alpha, wzp, ct, torch.empty(1, Nn, device=dev)))
        layers.append(lw)
    xs = {2048: torch.randn(1, 2048, device=dev), 5632: torch.randn(1, 5632, device=dev)}
    a8 = {k: torch.empty(1, k, dtype=torch.int8, device=dev) for k in xs}
    rs = {k: torch.empty(1, dtype=torch.int32, device=dev) for k in xs}
    from mobilequant_amd import _lib
    st = lambda: torch.cuda.current_stream().cuda_stream

    def token():
        for lw in layers:
            for (Kk, Nn), (w8, alpha, wzp, ct, out) in zip(shapes, lw):
                ops.int8_linear_f32in(xs[Kk], aq.scale, aq.offset, 0.0, 255.0, 128, w8, alpha, wzp, ct, None, out_scale=oq.scale,
                                      out_offset=oq.offset, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_F32, out=out, w4=w4)
    t = event_time(token, 1)      # graph of one token, best of 5 replays
    wbytes = 22 * sum(Kk * Nn for Kk, Nn in shapes) // (2 if w4 else 1)
    return {"decode_tok_s": round(1.0 / t, 1), "ms_per_token": round(t * 1e3, 4), "weight_GB_per_token": round(wbytes / 1e9, 4),
            "achieved_GBps": round(wbytes / t / 1e9, 1), "peak_GBps": 8000.0, "kernels_per_token": 22 * 4,
            "scope": "linears only (22 layers x [qkv, o, w1|w3, w2] %s GEMV with the activation quantize fused in), batch 1, hipGraph" % ("W4A8" if w4 else "W8A8")}


PAIR_MODE_DEFAULT = 0      # mq_gemm_set_pair_mode's built-in (mobilequant_amd_tuning.h)


def bench_pair(step):
    """w1 and w3 of the FFN in ONE launch (mq_w8a8_linear_tiled_pair: 512 tiles, two per CU, same activation panel): the
    headline GEMM as the layer actually runs it.  Both halves use the headline problem's operands."""
    from mobilequant_amd import ops, _lib
    from mobilequant_amd._lib import MQ_U8
    if not step.tiled:
        return None
    half = dict(w=step.w8, alpha=step.alpha, w_zp=step.wzp, col_term=step.ct, bias=None, out_scale=step.oq.scale, out_offset=step.oq.offset)
    step.quantize(0)
    t = event_time(lambda: ops.int8_linear_pair(step.a8s[0], M, step.rss[0], half, half, out_dtype=MQ_U8), 30)
    tops = 2 * OPS_PER_STEP / t / 1e12
    out = {"avg_launch_us": round(t * 1e6, 2), "us_per_gemm": round(t * 1e6 / 2, 2), "achieved_TOPS": round(tops, 1),
           "frac_of_int8_peak": round(tops / INT8_MFMA_PEAK_TOPS, 4),
           "note": "2 x (2048 x 2048 -> 5632) in one launch; allocates its two [M, N] outputs inside the timed call"}
    lib = _lib.load()
    if hasattr(lib, "mq_gemm_set_pair_mode"):          # A/B: one workgroup per tile runs both problems (tuning header)
        other = 1 - PAIR_MODE_DEFAULT
        lib.mq_gemm_set_pair_mode(other)
        try:
            t2 = event_time(lambda: ops.int8_linear_pair(step.a8s[0], M, step.rss[0], half, half, out_dtype=MQ_U8), 30)
        finally:
            lib.mq_gemm_set_pair_mode(PAIR_MODE_DEFAULT)
        out["persistent_over_the_pair_us" if other == 1 else "one_workgroup_per_tile_and_problem_us"] = round(t2 * 1e6, 2)
    return out


def bench_layer(dev):
    """The quantized-linear path of ONE TinyLlama decoder layer at prefill (S = 2048) through the module API:
    input_layernorm -> q/k/v, o_proj, post_attention_layernorm -> [w1, w3 -> act_fn -> * -> w2] (W8A8 recipe of
    ptq/mobilequant.py:175-201: 8-bit activations, 16-bit norm inputs, 16-bit o_proj / w2 outputs, per-channel w2).  Attention
    itself (RoPE, the two bmm's, softmax) and the residual adds are left out: q/k/v outputs are produced, o_proj gets a ready
    input.  Reports the hipGraph time with (a) the integer chain: fused norms -> int8 images -> GEMMs, the FFN as pair GEMM ->
    gated-activation kernel -> w2 GEMM (fuse_gated_mlp); (b) the chain of modules on their integer paths without the FFN fusion;
    (c) composite: fused_mode = "off", every linear quantising its own input."""
    import mobilequant_amd as mq
    from mobilequant_amd.quantization import qmodule as Q
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm
    a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
    S, H, F_, KV = 2048, 2048, 5632, 256
    torch.manual_seed(1337)

    def lin(k, n, own_input_quantizer, out_bits=8, per_channel=False):
        ql = mq.QLinear.from_float(torch.nn.Linear(k, n, bias=False).to(dev), a8, mq.QuantConfig(bitwidth=8, is_per_channel=per_channel),
                                   mq.QuantConfig(bitwidth=out_bits)).requires_grad_(False)
        if not own_input_quantizer:
            ql.input_quantizer = None
        ql.set_scale_offset({"input": [-4.0, 4.0], "output": [-3.0, 3.0]}, "buffer")
        return ql

    def norm():
        n = mq.QRMSNorm.from_float(HFRMSNorm(H, eps=1e-5).to(dev), a16, a16, a8).requires_grad_(False)
        n.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
        return n

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w1, self.w3, self.w2 = lin(H, F_, False), lin(H, F_, False), lin(F_, H, True, out_bits=16, per_channel=True)
            self.act_fn = mq.QSiLU(None, a8, a8)
            self.act_fn.set_scale_offset({"output": [-0.3, 3.0]}, "buffer")

        def forward(self, x):
            return self.w2(self.act_fn(self.w1(x)) * self.w3(x))

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.input_layernorm, self.post_attention_layernorm = norm(), norm()
            self.q_proj, self.k_proj, self.v_proj = lin(H, H, False), lin(H, KV, False), lin(H, KV, False)
            self.o_proj = lin(H, H, True, out_bits=16)
            self.mlp = MLP()

        def forward(self, x, attn_out):
            h = self.input_layernorm(x)
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            o = self.o_proj(attn_out)
            return q, k, v, o, self.mlp(self.post_attention_layernorm(x))
    layer = Layer()
    mq.wire_integer_inputs(layer)
    mq.fuse_gated_mlp(layer)
    x, attn_out = torch.randn(1, S, H, device=dev), torch.randn(1, S, H, device=dev)
    res = {}
    for mode in ("integer_chain", "module_chain", "composite"):
        for m in layer.modules():
            if hasattr(m, "fused_mode"):
                m.fused_mode = "off" if mode == "composite" else "auto"
        layer.mlp.fused_mode = "auto" if mode == "integer_chain" else "off"

        def fwd():
            if mode == "composite":
                Q._shared_activation.clear()
            layer(x, attn_out)
        fwd()
        res[mode + "_us"] = round(event_time(fwd, 5) * 1e6, 1)
    ops_layer = 2.0 * S * (H * H * 2 + H * KV * 2 + H * F_ * 3)
    res["tops_integer_chain"] = round(ops_layer / (res["integer_chain_us"] * 1e-6) / 1e12, 1)
    res["frac_of_int8_peak"] = round(res["tops_integer_chain"] / INT8_MFMA_PEAK_TOPS, 4)
    res["scope"] = ("one TinyLlama layer minus the attention core, S = 2048, W8A8 recipe: 2 QRMSNorm + q/k/v/o + gated FFN (w1, w3, QSiLU, "
                    "product, w2), module API, hipGraph")
    return res


def bench_layer_full(dev, modes=("fused", "attention_chain", "composite"), wbits=8, family="tinyllama"):
    """ONE whole TinyLlama decoder layer at prefill (B = 1, S = 2048) on the reference's module graph (mobilequant_amd/llama.py:
    norms, q/k/v/o, RoPE, qk_bmm / pv_bmm QMatMuls, softmax, gated FFN, residual adds), W8A8 recipe of ptq/mobilequant.py:175-201,
    ranges from this package's own calibration pass over the fp32 layer.  hipGraph time with (a) everything fused (fuse_attention:
    integer q.k^T / p.v with the softmax in one kernel; fuse_gated_mlp; fused norms), (b) the attention as the chain of modules
    (the [32, S, S] score tensor goes through memory ~6 times), (c) composite: every fused_mode off."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.quantization import qmodule as Q
    S = 2048
    shape = getattr(llama.LlamaShape, family)(layers=1, max_pos=S, vocab=4096)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, shape.vocab, (1, S), generator=g).to(dev)
    with torch.no_grad():
        act = get_act_range(model, [ids])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(model, a8 if wbits == 8 else mq.QuantConfig(bitwidth=wbits, is_per_channel=True), a8)
    for name, mod in model.named_modules():                 # ptq/mobilequant.py:175-201
        if isinstance(mod, mq.QLinear) and "w2" in name:
            mod.weight_quantizer.qcfg.is_per_channel = True
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QLinear) and "o_proj" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, (mq.QRMSNorm, mq.QLayerNorm)):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul) and "qk_bmm" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul) and "pv_bmm" in name:
            mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "buffer")
    mq.wire_integer_inputs(model)
    llama.fuse_decoder_layer(model)          # fuse_attention + fuse_gated_mlp + residual adds inside the o_proj / w2 GEMM stores
    layer = model.layers[0]
    with torch.no_grad():
        x = model.embed_tokens(ids)
    cos, sin = model.cos[:S], model.sin[:S]
    mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
    mask._mq_causal = True
    res = {}
    outs = {}
    for mode in modes:
        for m in layer.modules():
            if hasattr(m, "fused_mode"):
                m.fused_mode = "off" if mode == "composite" else "auto"
        layer.mlp.fused_mode = "off" if mode == "composite" else "auto"
        layer.self_attn.fused_mode = "auto" if mode == "fused" else "off"
        layer.fused_mode = "off" if mode == "composite" else "auto"

        def fwd():
            if mode == "composite":
                Q._shared_activation.clear()
            with torch.no_grad():
                outs[mode] = layer(x, cos, sin, mask)
        fwd()
        res[mode + "_us"] = round(event_time(fwd, 3) * 1e6, 1)
    # the attention op alone (prep + core kernels), with the arguments the fused layer hands it
    from mobilequant_amd import ops as _ops
    rec, real = [], _ops.attention_quant
    _ops.attention_quant = lambda *a, **k: (rec.append((a, k)), real(*a, **k))[1]
    try:
        layer.self_attn.fused_mode, layer.mlp.fused_mode, layer.fused_mode = "auto", "auto", "auto"
        with torch.no_grad():
            layer(x, cos, sin, mask)
    finally:
        _ops.attention_quant = real
    if rec:                                    # (head_dim != 64: the attention runs as its module chain, no fused op to time)
        (a_args, a_kw), = rec
        res["attention_op_us"] = round(event_time(lambda: real(*a_args, **a_kw), 5) * 1e6, 1)
        if shape.head_dim == 64:                 # the same op with the int8 score contraction (round 4's form; identical results)
            import mobilequant_amd._lib as _L
            _L.load().mq_attention_set_f16(0)
            try:
                res["attention_op_int8_scores_us"] = round(event_time(lambda: real(*a_args, **a_kw), 5) * 1e6, 1)
            finally:
                _L.load().mq_attention_set_f16(1)
    else:
        res["attention_op_us"] = None
    if "fused" in outs and "attention_chain" in outs:
        span = float(outs["attention_chain"].max() - outs["attention_chain"].min())
        res["fused_vs_chain_max_over_span"] = round(float((outs["fused"] - outs["attention_chain"]).abs().max()) / span, 5)
        res["fused_vs_chain_median_over_span"] = round(float((outs["fused"] - outs["attention_chain"]).abs().median()) / span, 8)
        res["parity_note"] = ("a position whose 8-bit index flips upstream moves by ~1-2 % of the span in either path; against the reference's own "
                              "logits the fused layer and the module chain are equally close (tests/golden/layer_case.npz: max 1.8 %, median 6e-7)")
    hidden, kv, ffn = shape.hidden, shape.kv_heads * shape.head_dim, shape.ffn
    ops_lin = 2.0 * S * (hidden * hidden * 2 + hidden * kv * 2 + hidden * ffn * 3)
    ops_att = 2.0 * shape.heads * shape.head_dim * S * S          # causal: q.k^T + p.v, each 2 * S^2 / 2 * D per head
    if "fused_us" in res:
        res["tops_fused"] = round((ops_lin + ops_att) / (res["fused_us"] * 1e-6) / 1e12, 1)
        res["frac_of_int8_peak"] = round(res["tops_fused"] / INT8_MFMA_PEAK_TOPS, 4)
        if rec:
            res["launches_per_layer"] = 9      # 4-bit weights run the same int8 kernels on their one-byte-per-nibble image
    res["scope"] = f"one whole {family} decoder layer, B = 1, S = 2048, W{wbits}A8 recipe, module API, hipGraph"
    return res


def bench_train_step(dev, S=2048):
    """SURVEY 8(f) rank 3, measured: ONE inner step of e2equant (algorithm.py:692-760 under the deployment recipe's flags --lwc --let
    --lrl --deactive_amp: fp32, 4-bit per-channel weights, learnable activation ranges, LET scales, LWC bound factors) on one
    TinyLlama-shaped decoder layer at S tokens: smooth_lm_temporary -> quantized forward -> MSE against the fp layer's output ->
    backward.  Every Quantizer.forward / backward in it is a HIP pass (STE, clamp mask, LSQ gradients): the per-tensor / per-row
    fake-quant pair, the LWC weight grids as mq_lwc_fake_quant (range + bound factors + grid + fake-quant in one pass per direction)
    and the score-sized chain of the attention block as mq_attention_probs_train; the GEMMs are torch's fp32 library kernels, as in
    the reference.  (Parity of exactly this step: tests/golden/train_step.npz.)"""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import get_act_range
    shape = llama.LlamaShape.tinyllama(layers=1, max_pos=S, vocab=4096)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337, std=0.05)
    for lin in (m for m in model.layers[0].modules() if isinstance(m, torch.nn.Linear)):
        lin.bias = torch.nn.Parameter(torch.zeros(lin.out_features))
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, shape.vocab, (1, S), generator=g).to(dev)
    cos, sin = model.cos[:S], model.sin[:S]
    mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
    with torch.no_grad():
        act = get_act_range(model, [ids])
        x = model.embed_tokens(ids)
        y_fp = model.layers[0](x, cos, sin, mask)
    mq.create_sim_qmodel(model, mq.QuantConfig(bitwidth=4, is_per_channel=True), mq.QuantConfig(bitwidth=8))
    layer = model.layers[0]
    for name, mod in layer.named_modules():                 # ptq/mobilequant.py:175-201
        if isinstance(mod, mq.QLinear) and "w2" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QLinear) and "o_proj" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
        elif isinstance(mod, mq.QMatMul) and "qk_bmm" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul) and "pv_bmm" in name:
            mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "parameter")        # learnable ranges (--lrl)
    for mod in layer.modules():
        if isinstance(mod, (mq.QLinear, mq.QRMSNorm)):
            mod.weight_quantizer.enable_lwc(mod.weight)
    for name, width in (("qkv", shape.hidden), ("fc1", shape.hidden), ("out", shape.heads * shape.head_dim), ("fc2", shape.ffn)):   # in_features of the
        # consumer linear, as algorithm.py:699-706 registers them (the v -> o pair itself only applies without GQA, algorithm.py:212)
        layer.register_parameter(f"{name}_smooth_scale", torch.nn.Parameter(torch.ones(width, device=dev)))
        layer.register_parameter(f"{name}_smooth_shift", torch.nn.Parameter(torch.zeros(width, device=dev)))
    train = [p for n, p in layer.named_parameters() if any(t in n for t in ("bound_factor", "smooth_scale", "quantizer.scale", "quantizer.offset"))]
    for p in layer.parameters():
        p.requires_grad_(False)
    for p in train:
        p.requires_grad_(True)
    cfg = type("Cfg", (), dict(shared_attention_norm=False, num_linears_per_mlp=3))()
    loss_fn = torch.nn.MSELoss()

    def step():
        for p in train:
            p.grad = None
        with torch.enable_grad():                           # (the variant legs run under no_grad)
            mq.smooth_lm_temporary(layer, cfg, True, False)
            loss = loss_fn(y_fp, layer(x, cos, sin, mask))
            loss.backward()
        return loss
    torch.cuda.reset_peak_memory_stats(dev)
    loss0 = float(step().detach())
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    peak = torch.cuda.max_memory_allocated(dev) / 1e9
    grads = sum(1 for p in train if p.grad is not None and torch.isfinite(p.grad).all())
    return {"ms_per_step": round(1e3 * sorted(times)[len(times) // 2], 2), "tokens": S, "loss": round(loss0, 6), "trainable_tensors": len(train),
            "tensors_with_finite_grad": grads, "peak_GB": round(peak, 2),
            "scope": f"one e2equant inner step (LET + LWC + learnable ranges, W4 per-channel / A8, fp32) on one TinyLlama-shaped layer, S = {S}, eager"}


def bench_model_prefill(dev):
    """The whole simulated-quant forward the reference's eval / PTQ loops run (harness_eval, ptq/mobilequant.py): TinyLlama-1.1B shape,
    22 layers, vocab 32000, ONE 2048-token sequence, W8A8 recipe, ranges from this package's calibration pass over the fp32 model
    (random-init weights).  tokens/s with (a) llama.fuse_decoder_layer (9 launches per layer) and (b) every fused mode off (the HIP
    fake-quant kernels around library GEMMs, module by module).  embed_tokens, the final norm and the fp32 lm_head over all 2048
    positions are inside the timed region in both."""
    import mobilequant_amd as mq
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.quantization import qmodule as Q
    S = 2048
    shape = llama.LlamaShape.tinyllama(max_pos=S)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337, std=0.03)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, shape.vocab, (1, S), generator=g).to(dev)
    with torch.no_grad():
        act = get_act_range(model, [ids])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(model, a8, a8)
    for name, mod in model.named_modules():                 # ptq/mobilequant.py:175-201
        if isinstance(mod, mq.QLinear) and "w2" in name:
            mod.weight_quantizer.qcfg.is_per_channel = True
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QLinear) and "o_proj" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul) and "qk_bmm" in name:
            mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul) and "pv_bmm" in name:
            mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "buffer")
    mq.wire_integer_inputs(model)
    res = {}

    def set_mode(off):
        for m in model.modules():
            if hasattr(m, "fused_mode") or isinstance(m, (llama.DecoderLayer, llama.Attention, llama.MLP)):
                m.fused_mode = "off" if off else "auto"
            if isinstance(m, mq.QLinear):
                m.int8_mode = "off" if off else "auto"

    def fwd():
        Q._shared_activation.clear()
        with torch.no_grad():
            return model(ids)
    set_mode(True)
    fwd()
    t_sim = event_time(fwd, 1)
    set_mode(False)
    llama.fuse_decoder_layer(model)
    fwd()
    t_fused = event_time(fwd, 2)
    res["fused_ms"], res["simulated_ms"] = round(t_fused * 1e3, 3), round(t_sim * 1e3, 3)
    res["fused_tokens_per_s"], res["simulated_tokens_per_s"] = round(S / t_fused), round(S / t_sim)
    res["speedup"] = round(t_sim / t_fused, 2)

    def fwd_last():
        Q._shared_activation.clear()
        with torch.no_grad():
            return model(ids, last_logits_only=True)
    fwd_last()
    res["fused_last_logits_only_ms"] = round(event_time(fwd_last, 2) * 1e3, 3)     # the context encoding of generation (DecodeEngine.prefill)
    res["scope"] = ("TinyLlama-1.1B shape (22 layers, vocab 32000), one 2048-token sequence, W8A8 recipe, module API, hipGraph; 'simulated' = "
                    "the reference's execution model on this GPU (fake-quant kernels around fp32 library GEMMs, every module on its own)")
    del model
    torch.cuda.empty_cache()
    return res


def bench_other_configs(dev):
    """BASELINE.json configs[2] and configs[3] as module-API QLinear steps at M = 2048 (fp32 in -> quantize -> int8 GEMM -> 8-bit output
    indices' fp32 values), hipGraph, per shape of the model's FFN / attention linears:
      configs[2] stablelm-2-1.6B W8A8, per-channel weight grids (hidden 2048, FFN 5632);
      configs[3] gemma-2B W4A8, packed 4-bit weights unpacked in registers in front of the int8 MFMA (hidden 2048, FFN 16384)."""
    import mobilequant_amd as mq
    out = {}
    torch.manual_seed(7)
    x = torch.randn(1, 2048, 2048, device=dev)
    for cfg, wbits, per_channel, shapes in (("configs[2] stablelm-2-1.6B W8A8 per-channel", 8, True, (("w1/w3", 2048, 5632), ("w2", 5632, 2048), ("q/o", 2048, 2048))),
                                            ("configs[3] gemma-2B W4A8", 4, False, (("w1/w3", 2048, 16384), ("w2", 16384, 2048), ("q/o", 2048, 2048)))):
        rows = {}
        for name, k, n in shapes:
            lin = torch.nn.Linear(k, n, bias=False, device=dev)
            a8 = mq.QuantConfig(bitwidth=8)
            ql = mq.QLinear.from_float(lin, a8, mq.QuantConfig(bitwidth=wbits, is_per_channel=per_channel), a8).requires_grad_(False)
            xin = x if k == 2048 else torch.randn(1, 2048, k, device=dev)
            ql.input_quantizer.set_scale_offset_from_minmax(float(xin.min()), float(xin.max()), "buffer", dev)
            ql.output_quantizer.set_scale_offset_from_minmax(-3.0, 3.0, "buffer", dev)
            with torch.no_grad():
                ql(xin)
                t = event_time(lambda: ql(xin), 10)
            rows[f"{name} {n}<-{k}"] = {"us": round(t * 1e6, 1), "tops": round(2.0 * 2048 * k * n / t / 1e12, 1)}
            del ql, lin
            torch.cuda.empty_cache()
        out[cfg] = rows
    out["scope"] = ("QLinear.forward (quantize + int8 GEMM, fp32 out) per linear shape at M = 2048, hipGraph; the configs[3] rows above run the "
                    "4-bit weights as their one-byte-per-nibble int8 image (QLinear.w4_prefill = 'image', the default)")
    out["configs[3] packed 4-bit weights, generated ISA"] = bench_packed_w4(dev)
    return out


def bench_packed_w4(dev):
    """BASELINE.json configs[3] as it names it: PACKED 4-bit weights (mq_pack_w4: two nibbles per byte) -> int8 MFMA.  The GEMM alone on the
    index-output shapes the generated kernels serve (mq_w4a8_linear_tiled; fragment-blocked int8 activations, 8-bit output indices),
    next to the int8-image kernel on the SAME numbers: `expanded` = pieces split once per workgroup into the int8 W ring (frw4x, default),
    `per_wave` = every wave splits its own fragments in registers (frw4), `int8_image` = one byte per nibble on the int8 kernels."""
    import mobilequant_amd._lib as L
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_U8
    lib = L.load()
    res = {}
    for name, n, k in (("gemma w1/w3 16384<-2048", 16384, 2048), ("tinyllama w1/w3 5632<-2048", 5632, 2048), ("tinyllama q|k|v 2560<-2048", 2560, 2048)):
        g = torch.Generator().manual_seed(n)
        qw = torch.randint(0, 16, (n, k), generator=g, dtype=torch.uint8).to(dev)
        x = torch.randn(M, k, generator=g).to(dev)
        sc, of = torch.tensor([8.0 / 255], device=dev), torch.tensor([128.0], device=dev)
        a_t, rs = ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128)
        packed, w8 = ops.pack_w4(qw), qw.view(torch.int8)
        colsum = qw.to(torch.int32).sum(1).to(torch.int32)
        wsc = torch.rand(n, generator=g).to(dev) * 1e-2 + 1e-3
        wof = torch.randint(0, 16, (n,), generator=g).float().to(dev)
        alpha, wzp, ct = ops.linear_epilogue_prepare(sc, of, 128, wsc, wof, 0, colsum, k)
        so, oo = torch.tensor([0.05], device=dev), torch.tensor([128.0], device=dev)
        out = torch.empty(M, n, dtype=torch.uint8, device=dev)
        row = {}
        try:
            for key, mode in (("per_wave", 0), ("expanded", 1)):
                if lib.mq_gemm_set_w4_mode(mode) != 0:      # per-wave unpack: experiment builds only (build.py --experiments)
                    continue
                t = event_time(lambda: ops.w4a8_linear_tiled(a_t, M, packed, rs, alpha, wzp, ct, None, [(so, oo)], out=out), 20)
                row[key + "_us"] = round(t * 1e6, 2)
        finally:
            lib.mq_gemm_set_w4_mode(1)
        ref = out.clone()
        t8 = event_time(lambda: ops.int8_linear(a_t, w8, rs, alpha, wzp, ct, None, out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=255.0,
                                                out_dtype=MQ_U8, out=out, a_tiled_rows=M), 20)
        row["int8_image_us"] = round(t8 * 1e6, 2)
        row["identical_indices"] = bool(torch.equal(ref, out))
        row["frac_of_int8_peak_expanded"] = round(2.0 * M * n * k / (row["expanded_us"] * 1e-6) / 1e12 / INT8_MFMA_PEAK_TOPS, 4)
        row["weight_bytes"] = {"packed": n * k // 2, "int8_image": n * k}
        res[name] = row
        del qw, x, a_t, packed, w8, out, ref
        torch.cuda.empty_cache()
    res["kernel"] = "mq::gemm_i8_frw4_kernel<.., true> (tools/gen_fr_asm.py frw4x / frw4x_128) via mq_w4a8_linear_tiled"
    return res


def bench_variants(dev, step, args):
    """The non-headline legs of the qlinear workload (rank 0): module forward, pair GEMM, decode, layer benchmarks."""
    extras = {}
    # drop-in nn.Module forward: fp32 in -> fp32 out, weight cached as int8 after the first call
    import mobilequant_amd as mq
    lin = torch.nn.Linear(K, N, bias=False, device=dev)
    with torch.no_grad():
        lin.weight.copy_(step.w_fp)
    a8 = mq.QuantConfig(bitwidth=8)
    ql = mq.QLinear.from_float(lin, a8, a8, a8).requires_grad_(False)
    ql.input_quantizer.set_scale_offset_from_minmax(float(step.x[0].min()), float(step.x[0].max()), "buffer", dev)
    ql.output_quantizer.set_scale_offset_from_minmax(-3.0, 3.0, "buffer", dev)
    x3 = step.x[0].view(1, M, K)
    ql(x3)
    tm = event_time(lambda: ql(x3), 30)
    extras["module_forward_f32"] = {"ms_per_step": round(tm * 1e3, 5), "value": round(OPS_PER_STEP / tm / 1e12, 1),
                                    "note": "QLinear.forward from Python, eager (includes host launch overhead)"}
    extras["ffn_pair_gemm"] = bench_pair(step)
    decode = {}
    decode["linears_only_w8a8"] = bench_decode_linears(dev)
    torch.cuda.empty_cache()
    decode["linears_only_w4a8"] = bench_decode_linears(dev, w4=True)      # the reference's deployment mode: 4-bit weights
    torch.cuda.empty_cache()
    extras["layer_prefill"] = bench_layer(dev)
    extras["layer_prefill_full"] = bench_layer_full(dev)
    torch.cuda.empty_cache()
    extras["layer_prefill_full_w4a8"] = bench_layer_full(dev, modes=("fused", "composite"), wbits=4)     # 4-bit per-channel weights (int8 image at prefill)
    import mobilequant_amd as _mq
    _mq.QLinear.w4_prefill = "packed"          # ONE packed image per module (0.5 B / weight): q | k | v and w1 / w3 on mq_w4a8_linear_tiled
    try:
        r = bench_layer_full(dev, modes=("fused",), wbits=4)
        extras["layer_prefill_full_w4a8_packed_only"] = {"fused_us": r.get("fused_us"), "note": "QLinear.w4_prefill = 'packed': every 4-bit module "
                                                         "holds only the mq_pack_w4 image; q | k | v and w1 / w3 run the generated packed kernels, "
                                                         "o_proj / w2 the generated packed residual kernel (mq_w4a8_linear_tiled_residual)"}
    finally:
        _mq.QLinear.w4_prefill = "image"
    # BASELINE.json configs[2] / [3] on their own leaf graphs (LayerNorm + biased q|k|v + 25 % rotary; head_dim 256 / MQA / GeGLU / FFN 16384)
    extras["layer_prefill_full_stablelm_2_1_6b"] = bench_layer_full(dev, modes=("fused", "composite"), wbits=8, family="stablelm_2_1_6b")
    extras["layer_prefill_full_gemma_2b_w4a8"] = bench_layer_full(dev, modes=("fused", "composite"), wbits=4, family="gemma_2b")
    extras["train_step_e2equant"] = bench_train_step(dev)
    torch.cuda.empty_cache()
    torch.cuda.empty_cache()
    extras["model_prefill"] = bench_model_prefill(dev)
    torch.cuda.empty_cache()
    extras["calibration_reductions"] = bench_minmax(dev, 2048)      # HBM GB/s of the min / max kernels
    extras["decode_linears_only"] = decode
    return extras




if __name__ == "__main__":
    sys.argv.insert(1, "--variants")
    _b.main()
