#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (torch CPU) in the build container.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python oracle/gen_golden.py

The reference's Python cannot travel to the GPU box, so its outputs are frozen here as small data
fixtures (inputs + expected outputs) and committed with this script.  Nothing under
/root/reference is modified; three attributes that transformers 5.x dropped are restored on the
transformers modules in this process before the import.

Fixtures written (all data, no reference source text):
  scale_offset_grid.npz   a1/a2  compute_scale_offset_from_min_max / inverse over a grid
  quantizer_cases.npz     a5     Quantizer.forward outputs + integer indices, all qcfg combos
  quantizer_grads.npz     a5     autograd of Quantizer.forward: grad x / scale / offset (STE + clamp mask)
  qlinear_cases.npz       a8     QLinear.forward on small shapes (W8A8 / W4A8 / per-channel / bias)
  qlinear_grouped_cases.npz  a8  QLinear.forward with per-group weight grids (group_size 64 / 128 / 256), the per-group scale / offset
  calib_stream.npz        a12/a13 real get_act_range / get_act_scales on a toy module stack
  checksums.json          sha256 of full-size index tensors (inputs re-creatable from numpy seeds)
  qrmsnorm_cases.npz      a10    QRMSNorm.forward (16-bit input / weight grids, 8- or 16-bit output, mixed-precision rules)
  qact_cases.npz          a10    QSiLU / QGELU.forward (sigmoid grid [0,1], 8- and 16-bit outputs)
  nonfinite_cases.npz     a1/a3/a5 NaN and +-inf inputs: torch.clamp / amin / amax propagate NaN
  api_surface.json        state_dict keys / export_qcfg / export_act_range of a toy sim model
  lwc_cases.npz           a7     learnable weight clipping: forward values, gradients to the bound factors and the weight, run_lwc
  qmatmul_cases.npz       a10    QMatMul.forward at attention shapes (qk_bmm / pv_bmm mixed-precision rules)
  toy_lm_nll.npz          perplexity proxy: the reference's W8A8-sim logits + NLL of the toy LM on 96 tokens
  decode_case_stablelm.npz / decode_case_gemma.npz   configs[2] / [3] leaf graphs (LayerNorm + q|k|v bias + partial rotary; head_dim 256
                          + GeGLU + scaled embeddings) of the reference's HFForCausalLM at toy size, their recipes' logits
  train_step.npz          f3: one e2equant inner step (LET + LWC + LRL, fp32) on a decoder layer: loss and every trainable tensor's gradient
  generate_case.npz       f2: greedy free-running continuation (SimModel.generate's loop) of the decode_case model under the reference
  decode_case.npz         f2: W8A8-sim logits of the reference's real HFForCausalLM (2 layers) at every position of a sequence
  smooth_cases.npz        n1/f3/f4 on the reference's real HFForCausalLM (2 layers): fp logits, get_act_scales, smooth_lm fold,
                          smooth_lm_temporary / _inplace (LET) temp weights, Quantizer indices of x / s
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("MQ_REFERENCE", "/root/reference")
OUT = os.environ.get("MQ_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")   # (tools/check_golden.py regenerates into a temp dir)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

# --- harness-side shim for transformers drift (nothing in the reference is edited) -------------
import transformers.activations as ta            # noqa: E402
import transformers.cache_utils as cu            # noqa: E402
import transformers.utils.import_utils as iu     # noqa: E402

cu.SinkCache = type("SinkCache", (cu.Cache,), {})
iu.is_torch_fx_available = lambda: False
ta.ACT2FN["silu"] = nn.SiLU
for _n in ("lm_eval", "lm_eval.base", "lm_eval.evaluator", "lm_eval.tasks", "lm_eval.utils", "termcolor"):
    sys.modules.setdefault(_n, types.ModuleType(_n))
sys.modules["lm_eval.base"].BaseLM = object
sys.modules["termcolor"].colored = lambda s, *a, **k: s
for _k in ("base", "evaluator", "tasks", "utils"):
    setattr(sys.modules["lm_eval"], _k, sys.modules["lm_eval." + _k])

import mobilellm.quantization.qmodule as Q       # noqa: E402
from mobilellm.model.ops import FMatMul          # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(1)          # fixed summation order for the fp32 matmuls we freeze


def npf(t):
    return t.detach().cpu().numpy().copy()       # a COPY: .numpy() aliases the tensor, and smooth_lm / *_inplace rewrite weights in place


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------------
def gen_scale_offset_grid():
    mins = [-1.0, -3.5, 0.0, 0.25, 1.0, 2.0, -1e-7, -120.0, -0.013, 5.0, -7.25, -1e7, 0.0]
    maxs = [2.0, 3.5, 1.0, 0.75, 1.0, 3.0, 1e-7, 250.0, 0.021, 5.5, -0.5, 1e7, 0.0]
    rows = []
    for mn, mx in zip(mins, maxs):
        for bits in (4, 8, 16):
            for sym in (False, True):
                s, o, _, _, qmin, qmax = Q.compute_scale_offset_from_min_max(mn, mx, bits, sym)
                imn, imx = Q.compute_min_max_from_scale_offset(s, o, bits, sym)
                rows.append((mn, mx, bits, int(sym), float(s), float(o), qmin, qmax, float(imn), float(imx)))
    arr = np.array(rows, dtype=np.float64)
    # tensor (per-channel) form
    g = torch.Generator().manual_seed(7)
    tmn = -torch.rand(37, 1, generator=g) * 3
    tmx = torch.rand(37, 1, generator=g) * 2 + 0.01
    out = {"grid": arr, "tmin": npf(tmn), "tmax": npf(tmx)}
    for bits in (4, 8, 16):
        for sym in (False, True):
            s, o, *_ = Q.compute_scale_offset_from_min_max(tmn, tmx, bits, sym)
            out[f"t_scale_b{bits}_s{int(sym)}"] = npf(s)
            out[f"t_offset_b{bits}_s{int(sym)}"] = npf(o)
    np.savez_compressed(os.path.join(OUT, "scale_offset_grid.npz"), **out)


# ------------------------------------------------------------------------------------------------
def ref_index(qz, x):
    """Integer index the reference computes inside Quantizer.forward (qmodule.py:286-287),
    evaluated with the reference's own round_ste and the quantizer's cached state."""
    xx = x
    if qz.qcfg.is_per_channel and qz.qcfg.group_size != -1:
        xx = x.reshape(-1, qz.qcfg.group_size)
    q = (Q.round_ste(xx / qz.scale) + qz.offset).clamp(qz.qmin, qz.qmax)
    chk = ((q - qz.offset) * qz.scale).reshape(x.shape).type(x.dtype)
    return q.reshape(x.shape), chk


def special_values(shape, g):
    x = torch.randn(shape, generator=g) * 1.7
    flat = x.view(-1)
    flat[0], flat[1], flat[2], flat[3] = 0.0, -0.0, 1e-9, -1e-9
    flat[4], flat[5] = 40.0, -40.0            # far out of range
    return x


def gen_quantizer_cases():
    g = torch.Generator().manual_seed(1337)
    out, meta = {}, []
    cid = 0

    def run(x, bits, sym, per_ch, group, dynamic, rng, tag):
        nonlocal cid
        qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, group_size=group, is_symmetric=sym,
                                       is_per_channel=per_ch, is_dynamic=dynamic))
        if rng is not None:
            qz.set_scale_offset_from_minmax(rng[0], rng[1], "buffer", x.device)
        y = qz(x)
        if bits > 16:
            assert y is x
            return
        q, chk = ref_index(qz, x)
        assert torch.equal(chk, y), tag
        k = f"c{cid}"
        out[k + "_x"], out[k + "_y"], out[k + "_q"] = npf(x), npf(y), npf(q.float())
        out[k + "_scale"], out[k + "_offset"] = npf(qz.scale.float()), npf(qz.offset.float())
        rng_j = None
        if rng is not None:
            if torch.is_tensor(rng[0]):
                out[k + "_rmin"], out[k + "_rmax"] = npf(rng[0]), npf(rng[1])
                rng_j = "tensor"
            else:
                rng_j = list(rng)
        meta.append(dict(id=k, tag=tag, bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch,
                         group_size=group, is_dynamic=dynamic, rng=rng_j, qmin=qz.qmin, qmax=qz.qmax,
                         dtype=str(x.dtype).replace("torch.", "")))
        cid += 1

    for bits in (4, 8, 16):
        for sym in (False, True):
            x = special_values((6, 64), g)
            run(x, bits, sym, False, -1, False, (-2.5, 3.0), "static_per_tensor")
            run(x, bits, sym, False, -1, False, None, "first_forward_per_tensor")
            run(x, bits, sym, False, -1, True, None, "dynamic_per_tensor")
            run(x, bits, sym, True, -1, False, None, "per_channel")
            run(x, bits, sym, True, 32, False, None, "per_group32")
    # exact .5 ties: x = (k + 0.5) * scale with a power-of-two scale so x/scale is exact
    ties = (torch.arange(-20, 20, dtype=torch.float32) + 0.5) * 0.25
    run(ties.view(4, 10), 8, False, False, -1, False, (-32.0, 31.75), "ties_pow2_scale")   # scale = 0.25
    run(ties.view(4, 10), 8, True, False, -1, False, (-31.75, 31.75), "ties_sym")
    # ranges not containing zero / degenerate (SURVEY 8a' items 2-3)
    x = torch.rand(5, 16, generator=g) * 4
    run(x, 8, False, False, -1, False, (2.0, 3.0), "min_gt_zero")
    run(x, 8, False, False, -1, False, (1.0, 1.0), "degenerate")
    run(x, 8, False, False, -1, False, (0.0, 1.0), "sigmoid_range")
    run(x - 6.0, 8, False, False, -1, False, (-5.0, -1.0), "max_lt_zero")
    run(x, 16, False, False, -1, False, (-3.0, 4.0), "static16")
    run(x, 32, False, False, -1, False, None, "bypass32")
    # activation-like 3-D tensor, realistic range
    x3 = torch.randn(1, 24, 96, generator=g) * 3
    run(x3, 8, False, False, -1, False, (float(x3.min()), float(x3.max())), "act3d_tensor_range")
    run(x3, 16, False, False, -1, False, (float(x3.min()), float(x3.max())), "act3d_16bit")
    # fp16 inputs: 0-dim scale keeps fp16 math, [N,1] scale promotes (SURVEY 8a' item 4)
    xh = special_values((6, 64), g).half()
    run(xh, 8, False, False, -1, False, (-2.5, 3.0), "f16_static_per_tensor")
    run(xh, 8, True, False, -1, False, (-2.5, 3.0), "f16_static_per_tensor_sym")
    # (fp16 first-forward range computation is impossible in the reference: clamp(max=1e6) overflows
    #  half, qmodule.py:58 -- so the per-channel fp16 case uses preset fp32 [N,1] ranges, which promote)
    run(xh, 8, False, True, -1, False, (-torch.rand(6, 1, generator=g) - 1.0, torch.rand(6, 1, generator=g) + 1.5),
        "f16_per_channel_preset")
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "quantizer_cases.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_nonfinite():
    """NaN / +-inf inputs through the reference: Quantizer.forward with a preset grid (torch.clamp keeps
    NaN, saturates inf), compute_min_max_from_tensor (amin/amax return NaN if any element is NaN) and
    compute_scale_offset_from_min_max on a NaN range."""
    g = torch.Generator().manual_seed(4242)
    out, meta = {}, []
    x = torch.randn(8, 64, generator=g) * 2
    x[0, 3], x[2, 0], x[7, 63] = float("nan"), float("nan"), float("nan")
    x[1, 5], x[3, 9], x[4, 4] = float("inf"), float("-inf"), float("inf")
    for i, (bits, sym, per_ch) in enumerate(((8, False, False), (8, True, False), (16, False, False), (8, False, True))):
        qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch))
        if per_ch:
            rng = (-torch.rand(8, 1, generator=g) - 1.0, torch.rand(8, 1, generator=g) + 1.5)
            out[f"q{i}_rmin"], out[f"q{i}_rmax"] = npf(rng[0]), npf(rng[1])
        else:
            rng = (-2.5, 3.0)
        qz.set_scale_offset_from_minmax(rng[0], rng[1], "buffer", x.device)
        out[f"q{i}_y"] = npf(qz(x))
        meta.append(dict(id=f"q{i}", bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch,
                         rng=None if per_ch else list(rng)))
    out["x"] = npf(x)
    # statistics: NaN rows / columns, inf-only rows, per-tensor
    xs = torch.randn(6, 40, generator=g)
    xs[1, 7] = float("nan")
    xs[3, 0], xs[4, 39] = float("inf"), float("-inf")
    out["xs"] = npf(xs)
    mn, mx = Q.compute_min_max_from_tensor(xs)
    out["xs_t_min"], out["xs_t_max"] = npf(mn), npf(mx)
    mn, mx = Q.compute_min_max_from_tensor(xs, is_per_channel=True)
    out["xs_r_min"], out["xs_r_max"] = npf(mn), npf(mx)
    out["xs_c_min"], out["xs_c_max"] = npf(torch.amin(xs, 0)), npf(torch.amax(xs, 0))       # generate_act_range.py:62-63
    xi = xs.clone()
    xi[1, 7] = 0.25                      # infinities only
    out["xi"] = npf(xi)
    mn, mx = Q.compute_min_max_from_tensor(xi)
    out["xi_t_min"], out["xi_t_max"] = npf(mn), npf(mx)
    # NaN / inf ranges -> scale, offset
    rmin = torch.tensor([float("nan"), -1.0, -1.0, float("-inf"), -2.0])
    rmax = torch.tensor([1.0, float("nan"), float("inf"), 1.0, 3.0])
    for sym in (False, True):
        sc, of, *_ = Q.compute_scale_offset_from_min_max(rmin, rmax, 8, sym)
        out[f"so_scale_s{int(sym)}"], out[f"so_offset_s{int(sym)}"] = npf(sc), npf(of)
    out["so_min"], out["so_max"] = npf(rmin), npf(rmax)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "nonfinite_cases.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_quantizer_grads():
    """Gradients the reference's autograd produces through Quantizer.forward (STE round, clamp mask,
    learnable scale / offset): what algorithm.py's LRL / LWC training consumes."""
    g = torch.Generator().manual_seed(99)
    out, meta = {}, []
    with torch.enable_grad():
        for cid, (bits, sym, per_ch, rng) in enumerate([(8, False, False, (-2.0, 2.5)), (8, True, False, (-2.0, 2.5)),
                                                        (4, False, False, (-1.0, 1.5)), (8, False, True, None),
                                                        (4, True, True, None), (16, False, False, (-3.0, 3.0))]):
            x = (torch.randn(12, 96, generator=g) * 1.5).requires_grad_(True)
            gy = torch.randn(12, 96, generator=g)
            qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch))
            if rng is not None:
                qz.set_scale_offset_from_minmax(rng[0], rng[1], "parameter")
            else:
                with torch.no_grad():
                    qz(x.detach())                      # first forward caches the per-row grid as Parameters
            y = qz(x)
            (y * gy).sum().backward()
            k = f"g{cid}"
            out[k + "_x"], out[k + "_gy"], out[k + "_gx"] = npf(x), npf(gy), npf(x.grad)
            out[k + "_scale"], out[k + "_offset"] = npf(qz.scale), npf(qz.offset)
            out[k + "_gscale"], out[k + "_goffset"] = npf(qz.scale.grad), npf(qz.offset.grad)
            meta.append(dict(id=k, bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch, qmin=qz.qmin, qmax=qz.qmax))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "quantizer_grads.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_qrmsnorm_cases():
    """QRMSNorm.forward of the reference (qmodule.py:469-530) on small shapes, configured the way
    ptq/mobilequant.py:184-188 does (16-bit input and weight grids, asymmetric per-tensor weight), plus the
    variants without an input / output quantizer and with a 16-bit output."""
    from mobilellm.model.hf_model import HFRMSNorm
    g = torch.Generator().manual_seed(2024)
    out, meta = {}, []
    for cid, (rows, cols, in_bits, out_bits) in enumerate(((24, 256, 16, 8), (7, 2048, 16, 8), (24, 256, None, 8),
                                                           (24, 256, 16, 16), (24, 256, 16, None), (5, 5632, 16, 8))):
        fp = HFRMSNorm(cols, eps=1e-5)
        with torch.no_grad():
            fp.weight.copy_(torch.randn(cols, generator=g) * 0.3 + 1.0)
        a16 = Q.QuantConfig(bitwidth=16)
        qn = Q.QRMSNorm.from_float(fp, Q.QuantConfig(bitwidth=in_bits) if in_bits else None, a16,
                                   Q.QuantConfig(bitwidth=out_bits) if out_bits else None)
        x = torch.randn(1, rows, cols, generator=g) * 2.5
        x[0, 0, :8] = torch.tensor([0.0, -0.0, 1e-9, -1e-9, 30.0, -30.0, 0.5, -0.5])
        act = {"input": [float(x.min()) * 0.9, float(x.max()) * 0.9]}      # clips a little on purpose
        y_fp = fp(x)
        act["output"] = [float(y_fp.min()) * 0.95, float(y_fp.max()) * 0.95]
        qn.set_scale_offset(act, "buffer")
        y = qn(x)
        k = f"n{cid}"
        out[k + "_x"], out[k + "_w"], out[k + "_y"] = npf(x), npf(fp.weight), npf(y)
        out[k + "_wscale"], out[k + "_woffset"] = npf(qn.weight_quantizer.scale.float()), npf(qn.weight_quantizer.offset.float())
        meta.append(dict(id=k, rows=rows, cols=cols, in_bits=in_bits, out_bits=out_bits, eps=1e-5, act=act))
    # QLayerNorm (StableLM-2: LayerNorm with bias), same mixed-precision rules
    for rows, cols, in_bits, out_bits in ((24, 256, 16, 8), (6, 2048, 16, 8), (24, 256, None, 16), (24, 256, 16, None)):
        fp = nn.LayerNorm(cols, eps=1e-5)
        with torch.no_grad():
            fp.weight.copy_(torch.randn(cols, generator=g) * 0.3 + 1.0)
            fp.bias.copy_(torch.randn(cols, generator=g) * 0.1)
        a16 = Q.QuantConfig(bitwidth=16)
        qn = Q.QLayerNorm.from_float(fp, Q.QuantConfig(bitwidth=in_bits) if in_bits else None, a16,
                                     Q.QuantConfig(bitwidth=out_bits) if out_bits else None)
        x = torch.randn(1, rows, cols, generator=g) * 2.5 + 0.7
        act = {"input": [float(x.min()) * 0.9, float(x.max()) * 0.9]}
        y_fp = fp(x)
        act["output"] = [float(y_fp.min()) * 0.95, float(y_fp.max()) * 0.95]
        qn.set_scale_offset(act, "buffer")
        y = qn(x)
        k = f"n{len(meta)}"
        out[k + "_x"], out[k + "_w"], out[k + "_b"], out[k + "_y"] = npf(x), npf(fp.weight), npf(fp.bias), npf(y)
        out[k + "_wscale"], out[k + "_woffset"] = npf(qn.weight_quantizer.scale.float()), npf(qn.weight_quantizer.offset.float())
        meta.append(dict(id=k, rows=rows, cols=cols, in_bits=in_bits, out_bits=out_bits, eps=1e-5, act=act, layernorm=True))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qrmsnorm_cases.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_qact_cases():
    """QSiLU.forward / QGELU.forward of the reference (qmodule.py:739-754, :790-798): no input quantizer (the surgery
    rule, mobilequant.py:196-199) and a 16-bit one; sigmoid grid = the [0,1] default or a calibrated input2 range."""
    g = torch.Generator().manual_seed(77)
    out, meta = {}, []
    x = torch.randn(1, 40, 352, generator=g) * 3
    x.view(-1)[:8] = torch.tensor([0.0, -0.0, 1e-9, -1e-9, 25.0, -25.0, 0.5, -0.5])
    out["x"] = npf(x)
    cid = 0
    for kind in ("silu", "gelu"):
        for in_bits, out_bits, mid in ((None, 8, None), (16, 8, [0.0, 0.98]), (None, 16, None), (None, None, None)):
            a = lambda b: Q.QuantConfig(bitwidth=b) if b else None      # noqa: E731
            if kind == "silu":
                m = Q.QSiLU(a(in_bits), Q.QuantConfig(bitwidth=8), a(out_bits))
                y_fp = torch.nn.functional.silu(x)
            else:
                m = Q.QGELU(a(in_bits), a(out_bits))
                y_fp = torch.nn.functional.gelu(x)
            act = {"input": [float(x.min()) * 0.9, float(x.max()) * 0.9], "output": [float(y_fp.min()), float(y_fp.max()) * 0.95]}
            if mid is not None and kind == "silu":
                act["input2"] = mid
            m.set_scale_offset(act, "buffer")
            k = f"a{cid}"
            out[k + "_y"] = npf(m(x))
            meta.append(dict(id=k, kind=kind, in_bits=in_bits, out_bits=out_bits, act=act))
            cid += 1
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qact_cases.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_qlinear_cases():
    g = torch.Generator().manual_seed(2024)
    out, meta = {}, []
    cid = 0

    def run(M, K, N, wbits, wsym, wpc, in_cfg, out_bits, bias, tag, x_on_grid=None):
        nonlocal cid
        lin = nn.Linear(K, N, bias=bias)
        lin.weight.copy_(torch.randn(N, K, generator=g) * 0.05)
        if bias:
            lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
        x = torch.randn(2, M // 2, K, generator=g) * 1.5
        wq = Q.QuantConfig(bitwidth=wbits, is_symmetric=wsym, is_per_channel=wpc)
        iq = Q.QuantConfig(**in_cfg) if in_cfg is not None else None
        oq = Q.QuantConfig(bitwidth=out_bits)
        ql = Q.QLinear.from_float(lin, iq if iq is not None else Q.QuantConfig(), wq, oq)
        if iq is None:
            ql.input_quantizer = None
        if x_on_grid is not None:
            # the producer's output grid: x is a fake-quantised tensor (q/k/v/o/w1/w3 case)
            pz = Q.Quantizer(Q.QuantConfig(bitwidth=x_on_grid))
            pz.set_scale_offset_from_minmax(float(x.min()), float(x.max()), "buffer")
            x = pz(x)
            out[f"c{cid}_xscale"], out[f"c{cid}_xoffset"] = npf(pz.scale), npf(pz.offset)
        y_fp = nn.functional.linear(x, ql.weight, ql.bias)
        act = {"output": [float(y_fp.min()), float(y_fp.max())]}
        if iq is not None:
            act["input"] = [float(x.min()), float(x.max())]
        ql.set_scale_offset(act, "buffer")
        y = ql(x)
        k = f"c{cid}"
        out[k + "_x"], out[k + "_w"], out[k + "_y"] = npf(x), npf(ql.weight), npf(y)
        if bias:
            out[k + "_b"] = npf(ql.bias)
        out[k + "_wscale"], out[k + "_woffset"] = npf(ql.weight_quantizer.scale), npf(ql.weight_quantizer.offset)
        meta.append(dict(id=k, tag=tag, M=M, K=K, N=N, wbits=wbits, wsym=wsym, wpc=wpc, in_cfg=in_cfg,
                         out_bits=out_bits, bias=bias, act=act, x_on_grid=x_on_grid))
        cid += 1

    a8 = dict(bitwidth=8)
    run(16, 128, 64, 8, False, False, None, 8, False, "w8a8_pt_noinq", x_on_grid=8)
    run(16, 128, 64, 8, False, False, a8, 8, False, "w8a8_pt_inq")
    run(16, 256, 96, 8, False, True, a8, 16, False, "w2_like_perch_out16")
    run(16, 128, 64, 8, False, True, None, 8, True, "stablelm_qkv_bias_perch", x_on_grid=8)
    run(16, 128, 64, 8, True, True, a8, 8, False, "w8_sym_perch")
    run(16, 128, 64, 4, True, True, a8, 8, False, "w4a8_sym_perch")
    run(16, 128, 64, 4, False, True, a8, 8, False, "w4a8_asym_perch")
    run(16, 128, 64, 4, False, True, None, 16, True, "w4a8_asym_perch_out16_bias", x_on_grid=8)
    run(8, 64, 32, 8, False, False, dict(bitwidth=8, is_symmetric=True), 8, False, "a8_sym_in")
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qlinear_cases.npz"), **out)


def gen_qlinear_dynamic_cases():
    """QLinear.forward with DYNAMIC activation quantizers (qmodule.py:262-277: the range is the tensor's own min / max on every call;
    CLI: ptq/mobilequant.py:50,166,205, ptq/generate_qcfg.py:33,78): dynamic input, dynamic output, both (+ bias, per-channel weights),
    and a dynamic 16-bit output.  The integer path serves them with device-resident grids (round 4)."""
    g = torch.Generator().manual_seed(4242)
    out, meta = {}, []
    for cid, (M, K, N, wpc, in_dyn, out_dyn, out_bits, bias, tag) in enumerate((
            (32, 128, 64, False, True, False, 8, False, "dyn_in"),
            (32, 128, 64, False, False, True, 8, False, "dyn_out"),
            (48, 256, 96, True, True, True, 8, True, "dyn_in_out_bias_perch"),
            (32, 128, 64, False, True, True, 16, False, "dyn_in_out16"))):
        lin = nn.Linear(K, N, bias=bias)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(N, K, generator=g) * 0.05)
            if bias:
                lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
        x = torch.randn(2, M // 2, K, generator=g) * 1.5
        ql = Q.QLinear.from_float(lin, Q.QuantConfig(bitwidth=8, is_dynamic=in_dyn), Q.QuantConfig(bitwidth=8, is_per_channel=wpc),
                                  Q.QuantConfig(bitwidth=out_bits, is_dynamic=out_dyn))
        y_fp = nn.functional.linear(x, ql.weight, ql.bias)
        act = {}
        if not in_dyn:
            act["input"] = [float(x.min()), float(x.max())]
        if not out_dyn:
            act["output"] = [float(y_fp.min()), float(y_fp.max())]
        if act:
            if "input" not in act:          # set_scale_offset wants both keys: set the static side by hand
                ql.output_quantizer.set_scale_offset_from_minmax(*act["output"], "buffer")
            elif "output" not in act:
                ql.input_quantizer.set_scale_offset_from_minmax(*act["input"], "buffer")
            else:
                ql.set_scale_offset(act, "buffer")
        with torch.no_grad():
            y = ql(x)
        k = f"c{cid}"
        out[k + "_x"], out[k + "_w"], out[k + "_y"] = npf(x), npf(ql.weight), npf(y)
        if bias:
            out[k + "_b"] = npf(ql.bias)
        out[k + "_oscale"] = npf(ql.output_quantizer.scale)
        meta.append(dict(id=k, tag=tag, M=M, K=K, N=N, wpc=wpc, in_dyn=in_dyn, out_dyn=out_dyn, out_bits=out_bits, bias=bias, act=act))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qlinear_dynamic_cases.npz"), **out)
    print("qlinear_dynamic_cases:", [m["tag"] for m in meta])


def gen_qlinear_grouped_cases():
    """QLinear.forward with PER-GROUP weight grids (Quantizer.group_size != -1: qmodule.py:259-260, :292-293; CLI --group_size,
    ptq/mobilequant.py:41, :157): 4- and 8-bit, asymmetric and symmetric, group sizes 64 / 128 / 256, with and without bias.  The weight
    quantizer's per-group scale / offset the reference derived are kept too.  Served on the integer path by mq_w8a8_linear_grouped (round 4)."""
    g = torch.Generator().manual_seed(777)
    out, meta = {}, []
    for cid, (M, K, N, gs, wbits, sym, bias, tag) in enumerate((
            (48, 512, 128, 128, 4, False, True, "w4_g128_bias"),
            (40, 256, 256, 64, 8, False, False, "w8_g64"),
            (64, 1024, 128, 256, 4, True, True, "w4_g256_sym_bias"),
            (33, 384, 128, 128, 8, True, False, "w8_g128_sym_ragged"))):
        lin = nn.Linear(K, N, bias=bias)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(N, K, generator=g) * 0.05 * (1 + torch.rand(N, 1, generator=g)))
            if bias:
                lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
        x = torch.randn(1, M, K, generator=g) * 1.5
        ql = Q.QLinear.from_float(lin, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=wbits, is_per_channel=True, group_size=gs, is_symmetric=sym),
                                  Q.QuantConfig(bitwidth=8))
        y_fp = nn.functional.linear(x, ql.weight, ql.bias)
        act = {"input": [float(x.min()), float(x.max())], "output": [float(y_fp.min()), float(y_fp.max())]}
        ql.set_scale_offset(act, "buffer")
        with torch.no_grad():
            y = ql(x)
        k = f"c{cid}"
        out[k + "_x"], out[k + "_w"], out[k + "_y"] = npf(x), npf(ql.weight), npf(y)
        if bias:
            out[k + "_b"] = npf(ql.bias)
        out[k + "_wscale"], out[k + "_woffset"] = npf(ql.weight_quantizer.scale), npf(ql.weight_quantizer.offset)
        out[k + "_oscale"] = npf(ql.output_quantizer.scale)
        meta.append(dict(id=k, tag=tag, M=M, K=K, N=N, gs=gs, wbits=wbits, sym=sym, bias=bias, act=act))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qlinear_grouped_cases.npz"), **out)
    print("qlinear_grouped_cases:", [m["tag"] for m in meta])


# ------------------------------------------------------------------------------------------------
class _Toy(nn.Module):
    """Toy module stack whose leaves are the classes the reference's calibration hooks select."""

    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(50, 32)
        self.fc1 = nn.Linear(32, 48)
        self.act = nn.SiLU()
        self.fc2 = nn.Linear(48, 32)
        self.bmm = FMatMul()
        self.ln = nn.LayerNorm(32)

    def forward(self, ids):
        h = self.emb(ids)
        c = self.fc2(self.act(self.fc1(h)))
        s = self.bmm(c, c.transpose(-1, -2))
        return self.ln(c) + s.mean()


class _Tok:
    bos_token_id, vocab_size = 1, 50

    def __call__(self, line, return_tensors="pt", max_length=None, truncation=True):
        ids = torch.tensor([[int(t) for t in line.split()][:max_length]])
        return types.SimpleNamespace(input_ids=ids)


def _load_script(path, argv):
    old = sys.argv
    sys.argv = argv
    try:
        spec = importlib.util.spec_from_file_location("_ref_script_" + os.path.basename(path)[:-3], path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = old
    return mod


def gen_calib_stream():
    torch.manual_seed(11)
    toy = _Toy().eval()
    g = torch.Generator().manual_seed(5)

    def make_ds(lens):
        return [{"text": " ".join(str(int(t)) for t in torch.randint(2, 49, (n,), generator=g))} for n in lens]

    def record(dataset, prefix, out):
        """What every hooked leaf saw (our own hooks) = the fixture's tensor stream."""
        names = {m: n for n, m in toy.named_modules()}
        count = {}

        def rec(m, xx, yy):
            n = names[m]
            k = count.get(n, 0)
            count[n] = k + 1
            out[f"{prefix}|{n}|input|{k}"] = npf(xx[0])
            out[f"{prefix}|{n}|output|{k}"] = npf(yy)
            if isinstance(m, FMatMul):
                out[f"{prefix}|{n}|input2|{k}"] = npf(xx[1])

        hs = [m.register_forward_hook(rec) for n, m in toy.named_modules()
              if isinstance(m, (nn.Linear, nn.SiLU, nn.LayerNorm, FMatMul))]
        for d in dataset:
            toy(_Tok()(d["text"], max_length=64).input_ids)
        for h in hs:
            h.remove()

    out = {}
    ragged = make_ds([5, 9, 12, 7, 3, 10])     # per-tensor + absmax: ragged sequence lengths
    fixed = make_ds([8] * 5)                   # per-channel: the reference needs a fixed length
    record(ragged, "stream_pt", out)           # (S x S matmul outputs change channel count otherwise)
    record(fixed, "stream_pc", out)

    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False
    pt = rng_mod.get_act_range(toy, _Tok(), ragged, len(ragged), 64)
    rng_mod.args.per_channel = True
    pc = rng_mod.get_act_range(toy, _Tok(), fixed, len(fixed), 64)
    for name, fields in pt.items():
        for f, v in fields.items():
            out[f"pt|{name}|{f}"] = np.array(v, dtype=np.float64)
    for name, fields in pc.items():
        for f, v in fields.items():
            out[f"pc|{name}|{f}"] = npf(v)
    ss_mod = _load_script(os.path.join(REF, "ptq", "generate_act_scale_shift.py"), ["x", "--hf_path", "none"])
    sc = ss_mod.get_act_scales(toy, _Tok(), ragged, len(ragged), 64)
    for k, v in sc.items():
        out[f"absmax|{k}"] = npf(v)
    ss_mod.args.use_rand_samples = False
    sh = ss_mod.get_act_shifts(toy, _Tok(), ragged, len(ragged), 64)       # EMA 0.99 / 0.01 of (max + min) / 2, in sample order
    for k, v in sh.items():
        out[f"shift|{k}"] = npf(v)
    for k, v in toy.state_dict().items():
        out["toy|" + k] = npf(v)
    out["ids_pt"] = np.array(json.dumps([d["text"] for d in ragged]))
    out["ids_pc"] = np.array(json.dumps([d["text"] for d in fixed]))
    np.savez_compressed(os.path.join(OUT, "calib_stream.npz"), **out)


# ------------------------------------------------------------------------------------------------
def gen_checksums():
    """Full-size index tensors: inputs come from numpy seeds so the tests can re-create them."""
    res = {}
    rng = np.random.default_rng(1337)
    x = (rng.standard_normal((2048, 2048), dtype=np.float32) * 1.0).astype(np.float32)
    xt = torch.from_numpy(x)
    for bits, sym in ((8, False), (8, True), (16, False)):
        qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, is_symmetric=sym))
        qz.set_scale_offset_from_minmax(float(xt.min()), float(xt.max()), "buffer")
        y = qz(xt)
        q, _ = ref_index(qz, xt)
        res[f"act_2048x2048_b{bits}_s{int(sym)}"] = dict(
            seed=1337, shape=[2048, 2048], rng=[float(xt.min()), float(xt.max())],
            q_sha256=sha(npf(q).astype(np.int32)), y_sha256=sha(npf(y)))
    rng = np.random.default_rng(4242)
    w = (rng.standard_normal((5632, 2048), dtype=np.float32) * 0.02).astype(np.float32)
    wt = torch.from_numpy(w)
    for bits, sym, pc in ((8, False, False), (8, False, True), (4, True, True), (4, False, True)):
        qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, is_symmetric=sym, is_per_channel=pc))
        y = qz(wt)
        q, _ = ref_index(qz, wt)
        res[f"w_5632x2048_b{bits}_s{int(sym)}_pc{int(pc)}"] = dict(
            seed=4242, shape=[5632, 2048], std=0.02,
            q_sha256=sha(npf(q).astype(np.int32)), y_sha256=sha(npf(y)),
            scale_sha256=sha(npf(qz.scale)), offset_sha256=sha(npf(qz.offset)))
    with open(os.path.join(OUT, "checksums.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


# ------------------------------------------------------------------------------------------------
class _Block(nn.Module):
    """Leaf-module graph of one llama-style decoder block (names drive the surgery rules)."""

    def __init__(self, d=32, f=48):
        super().__init__()
        from mobilellm.model.hf_model import HFRMSNorm
        self.input_layernorm = HFRMSNorm(d)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(d, d, bias=False) for _ in range(4))
        self.qk_bmm, self.pv_bmm = FMatMul(), FMatMul()
        self.post_attention_layernorm = HFRMSNorm(d)
        self.w1, self.w3, self.w2 = nn.Linear(d, f, bias=False), nn.Linear(d, f, bias=False), nn.Linear(f, d, bias=False)
        self.act_fn = nn.SiLU()

    def forward(self, x):
        h = self.input_layernorm(x)
        q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
        s = self.qk_bmm(q, k.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
        a = self.pv_bmm(torch.softmax(s, dim=-1), v)
        x = x + self.o_proj(a)
        h = self.post_attention_layernorm(x)
        return x + self.w2(self.act_fn(self.w1(h)) * self.w3(h))


class _ToyLM(nn.Module):
    def __init__(self):
        super().__init__()
        from mobilellm.model.hf_model import HFRMSNorm
        self.layers = nn.ModuleList([_Block(), _Block()])
        self.norm = HFRMSNorm(32)
        self.lm_head = nn.Linear(32, 50, bias=False)

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return self.lm_head(self.norm(x))


def gen_api_surface():
    torch.manual_seed(3)
    m = _ToyLM().eval()
    x = torch.randn(1, 6, 32, generator=torch.Generator().manual_seed(9))
    fp = m(x)
    sd0 = {k: npf(v) for k, v in m.state_dict().items()}
    # calibrate per-tensor ranges of every leaf with the reference hook set
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _M(nn.Module):                      # get_act_range feeds ids; wrap to feed our float input
        def __init__(s, inner):
            super().__init__(); s.inner = inner
        def forward(s, ids):
            return s.inner(x)
    act = rng_mod.get_act_range(_M(m), _Tok(), [{"text": "1 2 3"}], 1, 8)
    act = {k[len("inner."):]: v for k, v in act.items()}
    wq, aq = Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8)
    Q.create_sim_qmodel(m, wq, aq)
    # mixed precision rules of ptq/mobilequant.py:175-201
    for name, mod in m.named_modules():
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    qcfg = Q.export_qcfg(m)
    Q.set_scale_and_offset(m, act, "buffer")
    yq = m(x)
    exported = Q.export_act_range(m)
    surf = dict(qcfg=qcfg, act_dict=act, state_dict_keys=sorted(m.state_dict().keys()),
                exported_act_range=exported,
                module_types={n: type(mm).__name__ for n, mm in m.named_modules() if n})
    with open(os.path.join(OUT, "api_surface.json"), "w") as f:
        json.dump(surf, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "toy_lm.npz"), x=npf(x), y_fp=npf(fp), y_w8a8=npf(yq),
                        **{"sd|" + k: v for k, v in sd0.items()})


# ------------------------------------------------------------------------------------------------
def gen_lwc_cases():
    """a7: learnable weight clipping (qmodule.py:133-185, :262-277).  Every recipe passes --lwc, so the forward values with
    non-trivial bound factors, the gradients to upbound_factor / lowbound_factor / the weight (autograd of the reference,
    incl. the amin / amax path into the extreme elements) and run_lwc's clamped weights are frozen."""
    out, meta = {}, []
    g = torch.Generator().manual_seed(77)
    for tag, bits, sym, per_ch, shape in (("pt8", 8, False, False, (24, 64)), ("pc8", 8, False, True, (24, 64)),
                                          ("pc4", 4, False, True, (16, 128)), ("pc4s", 4, True, True, (16, 128))):
        w = (torch.randn(shape, generator=g) * 0.05)
        qz = Q.Quantizer(Q.QuantConfig(bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch))
        with torch.enable_grad():
            wp = w.clone().requires_grad_(True)
            qz.enable_lwc(wp)
            up0 = (torch.randn(qz.upbound_factor.shape, generator=g) * 0.7 + 1.5)
            lo0 = (torch.randn(qz.lowbound_factor.shape, generator=g) * 0.7 + 1.5)
            qz.upbound_factor.data.copy_(up0)
            qz.lowbound_factor.data.copy_(lo0)
            y = qz(wp)
            gy = torch.randn(shape, generator=g)
            (y * gy).sum().backward()
        out[tag + "_w"], out[tag + "_up"], out[tag + "_lo"], out[tag + "_gy"] = npf(w), npf(up0), npf(lo0), npf(gy)
        out[tag + "_y"] = npf(y)
        out[tag + "_scale"], out[tag + "_offset"] = npf(qz.scale), npf(qz.offset)
        out[tag + "_g_up"], out[tag + "_g_lo"], out[tag + "_g_w"] = npf(qz.upbound_factor.grad), npf(qz.lowbound_factor.grad), npf(wp.grad)
        out[tag + "_clamped"] = npf(qz.run_lwc(w))           # also drops the LWC state (qmodule.py:176-180)
        assert not qz.lwc and not hasattr(qz, "upbound_factor")
        meta.append(dict(id=tag, bitwidth=bits, is_symmetric=sym, is_per_channel=per_ch))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "lwc_cases.npz"), **out)
    print("lwc_cases:", len(meta))


def gen_qmatmul_cases():
    """a10: QMatMul.forward (qmodule.py:453-466) at attention shapes under the mixed-precision rules of
    ptq/mobilequant.py:190-201: qk_bmm (8-bit q, 8-bit k^T, 16-bit scores) and pv_bmm (16-bit probabilities, 8-bit v, 8-bit
    output), plus a plain 8/8/8 case; k^T enters as a transposed view exactly as hf_model.py:513 passes it."""
    out, meta = {}, []
    g = torch.Generator().manual_seed(123)
    H, S, D = 4, 48, 32
    q = torch.randn(1, H, S, D, generator=g)
    k = torch.randn(1, H, S, D, generator=g)
    v = torch.randn(1, H, S, D, generator=g)
    probs = torch.softmax(torch.randn(1, H, S, S, generator=g) * 2, dim=-1)

    def run(tag, a, b, bits):
        m = Q.QMatMul(Q.QuantConfig(bitwidth=bits[0]), Q.QuantConfig(bitwidth=bits[1]), Q.QuantConfig(bitwidth=bits[2]))
        y_fp = torch.matmul(a, b)
        act = {"input": [a.min().item(), a.max().item()], "input2": [b.min().item(), b.max().item()],
               "output": [y_fp.min().item(), y_fp.max().item()]}
        m.set_scale_offset(act, "buffer")
        y = m(a, b)
        out[tag + "_a"], out[tag + "_b"], out[tag + "_y"] = npf(a), npf(b.contiguous()), npf(y)
        meta.append(dict(id=tag, bits=list(bits), act=act, b_transposed=(tag != "pv")))
    run("qk", q, k.transpose(2, 3), (8, 8, 16))
    run("pv", probs, v, (16, 8, 8))
    run("qk888", q, k.transpose(2, 3), (8, 8, 8))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "qmatmul_cases.npz"), **out)
    print("qmatmul_cases:", len(meta))


def gen_toy_lm_nll():
    """Proxy for "quantized perplexity within 0.05 of the reference" (no checkpoints / datasets offline): the reference's own
    W8A8-sim logits of the 2-block toy LM on a 96-token input, with seeded labels -- the test compares the NLL / perplexity of
    the HIP path against the NLL / perplexity of these frozen logits."""
    torch.manual_seed(3)
    m = _ToyLM().eval()
    sd0 = {k: npf(v) for k, v in m.state_dict().items()}
    gg = torch.Generator().manual_seed(21)
    x = torch.randn(1, 96, 32, generator=gg)
    labels = torch.randint(0, 50, (96,), generator=gg)
    fp = m(x)
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _M(nn.Module):
        def __init__(s, inner):
            super().__init__(); s.inner = inner
        def forward(s, ids):
            return s.inner(x)
    act = rng_mod.get_act_range(_M(m), _Tok(), [{"text": "1 2 3"}], 1, 8)
    act = {k[len("inner."):]: v for k, v in act.items()}
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8))
    for name, mod in m.named_modules():          # mixed precision rules of ptq/mobilequant.py:175-201
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    Q.set_scale_and_offset(m, act, "buffer")
    yq = m(x)
    nll = lambda lg: float(torch.nn.functional.cross_entropy(lg[0], labels))      # noqa: E731
    np.savez_compressed(os.path.join(OUT, "toy_lm_nll.npz"), x=npf(x), labels=npf(labels), y_fp=npf(fp), y_w8a8=npf(yq),
                        act=np.array(json.dumps(act)), nll_fp=np.float64(nll(fp)), nll_w8a8=np.float64(nll(yq)),
                        **{"sd|" + k: v for k, v in sd0.items()})
    print("toy_lm_nll: nll fp %.5f w8a8 %.5f" % (nll(fp), nll(yq)))


# ------------------------------------------------------------------------------------------------
def _tiny_hf(kv_heads, seed):
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    cfg = HFConfig(vocab_size=50, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=kv_heads, max_position_embeddings=64, hidden_act="silu", use_matmul_as_module=True)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    m = HFForCausalLM(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():                      # livelier statistics than the default init
            if p_.dim() >= 2:
                p_.normal_(0.0, 0.3)
            else:
                p_.uniform_(0.5, 1.5)
    return m, cfg


def gen_smooth_cases():
    """n1 / f3 / f4 on the reference's REAL model classes (HFForCausalLM, 2 layers, hidden 64):
      * fp logits (pins mobilequant_amd/llama.py against hf_model.py);
      * get_act_scales (generate_act_scale_shift.py:42-93) -> smooth_lm (smoothquant.py:109-139): weights after the fold, with
        grouped-query heads (v -> o fold skipped, as for TinyLlama) and with full heads (v -> o folded);
      * smooth_lm_temporary (algorithm.py:187-233) with LET scales / shifts on a sim-quantised layer: every temp_weight /
        temp_bias and the layer output;
      * the run-time activation form: reference Quantizer indices of x / s for the scales smooth_ln_fcs derives."""
    sq = _load_script(os.path.join(REF, "ptq", "smoothquant.py"), ["x", "--hf_path", "none"])
    ss = _load_script(os.path.join(REF, "ptq", "generate_act_scale_shift.py"), ["x", "--hf_path", "none"])
    import mobilellm.quantization.algorithm as A
    out, meta = {}, {}
    for tag, kv in (("gqa", 2), ("mha", 4)):
        m, cfg = _tiny_hf(kv, 5 + kv)
        g = torch.Generator().manual_seed(31 + kv)
        ids = [torch.randint(0, 50, (1, 24), generator=g) for _ in range(3)]
        out[tag + "_ids"] = np.stack([npf(i[0]) for i in ids])
        for k_, v_ in m.state_dict().items():
            out[f"{tag}|sd|{k_}"] = npf(v_)
        out[tag + "_logits_fp"] = npf(m(ids[0], use_cache=False).logits)

        class _Tk:
            bos_token_id, vocab_size = 1, 50
            def __call__(s_, line, return_tensors="pt", max_length=None, truncation=True):
                return types.SimpleNamespace(input_ids=ids[int(line)])
        ss.args.use_rand_samples = False
        _orig = m.forward
        m.forward = lambda x_, **kw: _orig(x_, use_cache=False)      # the scripts call model(ids); transformers 5 needs use_cache=False
        scales = ss.get_act_scales(m, _Tk(), [{"text": str(i)} for i in range(3)], 3, 64)
        m.forward = _orig
        for k_, v_ in scales.items():
            out[f"{tag}|scale|{k_}"] = npf(v_)
        sq.smooth_lm(m, scales, 0.5, False, False)
        for k_, v_ in m.state_dict().items():
            out[f"{tag}|smoothed|{k_}"] = npf(v_)
        out[tag + "_logits_smoothed"] = npf(m(ids[0], use_cache=False).logits)
        meta[tag] = dict(kv_heads=kv, alpha=0.5)

    # ---- LET temporary weights on a sim-quantised decoder layer (MHA so that every pair exists) -----------------------
    m, cfg = _tiny_hf(4, 77)
    g = torch.Generator().manual_seed(5)
    layer = m.model.layers[0]
    for k_, v_ in layer.state_dict().items():
        out[f"let|sd|{k_}"] = npf(v_)
    Q.create_sim_qmodel(layer, Q.QuantConfig(bitwidth=8, is_per_channel=True), Q.QuantConfig(bitwidth=8))
    with torch.no_grad():
        for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj, layer.mlp.w1,
                    layer.mlp.w2, layer.mlp.w3):            # LET on q/k needs biases (algorithm.py:89-96); give every linear one
            lin.bias = nn.Parameter(torch.randn(lin.out_features, generator=g) * 0.1)
    for k_, v_ in layer.state_dict().items():
        if k_.endswith(".bias"):
            out[f"let|sd|{k_}"] = npf(v_)
    params = {}
    for name, dim in (("qkv", 64), ("fc1", 64), ("out", 64), ("fc2", 96), ("qkt", 64)):
        sc = torch.rand(dim, generator=g) * 1.5 + 0.5
        layer.register_parameter(f"{name}_smooth_scale", nn.Parameter(sc))
        params[f"{name}_smooth_scale"] = sc
        if name != "qkt":
            sh = torch.randn(dim, generator=g) * 0.1
            layer.register_parameter(f"{name}_smooth_shift", nn.Parameter(sh))
            params[f"{name}_smooth_shift"] = sh
    for k_, v_ in params.items():
        out["let|param|" + k_] = npf(v_)
    A.smooth_lm_temporary(layer, cfg, True, use_shift=True)
    for name, mod in layer.named_modules():
        if isinstance(mod, (Q.QLinear, Q.QRMSNorm)):
            out[f"let|temp_weight|{name}"] = npf(mod.temp_weight)
            if getattr(mod, "temp_bias", None) is not None:
                out[f"let|temp_bias|{name}"] = npf(mod.temp_bias)
    A.smooth_lm_inplace(layer, cfg, True, use_shift=True)
    for k_, v_ in layer.state_dict().items():
        if k_.endswith("weight") or k_.endswith("bias"):
            out[f"let|inplace|{k_}"] = npf(v_)

    # ---- the run-time activation form: indices of x / s ---------------------------------------------------------------
    x = torch.randn(40, 128, generator=g) * torch.linspace(0.2, 6.0, 128)         # outlier channels, what SmoothQuant is for
    w = torch.randn(96, 128, generator=g) * 0.05
    act_scales = x.abs().max(0)[0]
    wsc = w.abs().max(0)[0].clamp(min=1e-5)
    sc = (act_scales.pow(0.5) / wsc.pow(0.5)).clamp(min=1e-5)                     # smoothquant.py:60-62
    xs = x / sc
    qz = Q.Quantizer(Q.QuantConfig(bitwidth=8))
    qz.set_scale_offset_from_minmax(xs.min().item(), xs.max().item(), "buffer", None)
    y = qz(xs)
    out["cs_x"], out["cs_scales"], out["cs_range"] = npf(x), npf(sc), np.array([xs.min().item(), xs.max().item()], np.float32)
    out["cs_y"], out["cs_index"] = npf(y), npf(ref_index(qz, xs)[0])
    out["cs_w"], out["cs_w_scaled"] = npf(w), npf(w * sc.view(1, -1))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "smooth_cases.npz"), **out)
    print("smooth_cases:", len(out), "arrays")


def gen_artifacts():
    """a14: files as the reference's own writers produce them -- act_dict.json through mobilellm.utils.io.json_save (io.py:34-36)
    from the toy model's calibrated ranges, act_scales.pth through torch.save of get_act_scales' dictionary
    (generate_act_scale_shift.py:170-175).  Data files only."""
    from mobilellm.utils.io import json_save
    act = json.load(open(os.path.join(OUT, "api_surface.json")))["act_dict"]
    json_save(os.path.join(OUT, "act_dict_ref.json"), act)
    z = np.load(os.path.join(OUT, "smooth_cases.npz"))
    scales = {k.split("|", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith("gqa|scale|")}
    torch.save(scales, os.path.join(OUT, "act_scales_ref.pth"))
    print("artifacts: act_dict_ref.json", os.path.getsize(os.path.join(OUT, "act_dict_ref.json")), "bytes; act_scales_ref.pth", len(scales), "tensors")


def gen_decode_case():
    """f2: a decode loop must reproduce, token by token, what the reference's W8A8-simulated model computes for the same
    sequence (causal attention: position t depends on tokens <= t only; static per-tensor grids).  The REAL reference classes:
    a 2-layer HFForCausalLM (hidden 256, 4 heads / 2 KV heads, head_dim 64, FFN 512 -- sizes the decode kernels accept),
    calibrated by the reference's get_act_range, create_sim_qmodel + the mixed-precision rules of ptq/mobilequant.py:175-201,
    logits at every position of a 40-token sequence."""
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    cfg = HFConfig(vocab_size=96, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=2, max_position_embeddings=64, hidden_act="silu", use_matmul_as_module=True)
    cfg._attn_implementation = "eager"
    torch.manual_seed(2024)
    m = HFForCausalLM(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() >= 2:
                p_.normal_(0.0, 0.08)
            else:
                p_.uniform_(0.7, 1.3)
    out = {"sd|" + k_: npf(v_) for k_, v_ in m.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 96, (1, 40), generator=g)
    calib = [torch.randint(0, 96, (1, 40), generator=g) for _ in range(4)] + [ids]
    out["ids"] = npf(ids[0])
    out["logits_fp"] = npf(m(ids, use_cache=False).logits)
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _Tk:
        bos_token_id, vocab_size = 1, 96
        def __call__(s_, line, return_tensors="pt", max_length=None, truncation=True):
            return types.SimpleNamespace(input_ids=calib[int(line)])
    _orig = m.forward
    m.forward = lambda x_, **kw: _orig(x_, use_cache=False)
    act = rng_mod.get_act_range(m, _Tk(), [{"text": str(i)} for i in range(len(calib))], len(calib), 64)
    m.forward = _orig
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8))
    for name, mod in m.named_modules():          # ptq/mobilequant.py:175-201
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    act = {k_: v_ for k_, v_ in act.items() if any(k_ == n for n, mm in m.named_modules() if isinstance(mm, (Q.QLinear, Q.QRMSNorm, Q.QMatMul, Q.QSiLU)))}
    Q.set_scale_and_offset(m, act, "buffer")
    out["logits_w8a8"] = npf(m(ids, use_cache=False).logits)
    out["act"] = np.array(json.dumps(act))
    out["qcfg"] = np.array(json.dumps(Q.export_qcfg(m)))
    np.savez_compressed(os.path.join(OUT, "decode_case.npz"), **out)
    d = out["logits_w8a8"] - out["logits_fp"]
    print("decode_case: logits", out["logits_w8a8"].shape, "max |w8a8 - fp| %.4f of span %.3f" % (np.abs(d).max(), np.ptp(out["logits_fp"])))


def gen_decode_case_w4(tag="w4", wbits=4, kv_heads=2, act="silu", heads=4, wsym=False, cfg_kw=None):
    """The reference's deployment recipe on the 2-layer model of gen_decode_case: packed-4-bit-style weights (4-bit per-channel
    asymmetric, as experiments/w4a8/main/e2e_llama-s1024-ep60.sh:23), 8-bit activations, mixed-precision rules of
    ptq/mobilequant.py:175-201.  Weights from tests/seeded.py (not stored); logits of the REAL HFForCausalLM at every position."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from seeded import seeded_parameters_
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    cfg = HFConfig(vocab_size=96, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=heads,
                   num_key_value_heads=kv_heads, max_position_embeddings=64, hidden_act=act, use_matmul_as_module=True, **(cfg_kw or {}))
    cfg._attn_implementation = "eager"
    m = HFForCausalLM(cfg).eval()
    seeded_parameters_(m, std=0.08, strip="model.")
    g = torch.Generator().manual_seed(18)
    ids = torch.randint(0, 96, (1, 40), generator=g)
    calib = [torch.randint(0, 96, (1, 40), generator=g) for _ in range(4)] + [ids]
    out = {"ids": npf(ids[0])}
    with torch.no_grad():
        out["logits_fp"] = npf(m(ids, use_cache=False).logits)
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _Tk:
        bos_token_id, vocab_size = 1, 96
        def __call__(s_, line, return_tensors="pt", max_length=None, truncation=True):
            return types.SimpleNamespace(input_ids=calib[int(line)])
    _orig = m.forward
    m.forward = lambda x_, **kw: _orig(x_, use_cache=False)
    act = rng_mod.get_act_range(m, _Tk(), [{"text": str(i)} for i in range(len(calib))], len(calib), 64)
    m.forward = _orig
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=wbits, is_per_channel=True, is_symmetric=wsym), Q.QuantConfig(bitwidth=8))
    for name, mod in m.named_modules():          # ptq/mobilequant.py:175-201
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, (Q.QRMSNorm, Q.QLayerNorm)):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    act = {k_: v_ for k_, v_ in act.items() if any(k_ == n for n, mm in m.named_modules()
                                                  if isinstance(mm, (Q.QLinear, Q.QRMSNorm, Q.QLayerNorm, Q.QMatMul, Q.QSiLU, Q.QGELU)))}
    Q.set_scale_and_offset(m, act, "buffer")
    with torch.no_grad():
        out["logits_w4a8"] = npf(m(ids, use_cache=False).logits)      # (key kept for every variant: the quantised logits)
    out["act"] = np.array(json.dumps(act))
    out["qcfg"] = np.array(json.dumps(Q.export_qcfg(m)))
    np.savez_compressed(os.path.join(OUT, f"decode_case_{tag}.npz"), **out)
    d = out["logits_w4a8"] - out["logits_fp"]
    print(f"decode_case_{tag}: logits", out["logits_w4a8"].shape, "max |quantised - fp| %.4f of span %.3f" % (np.abs(d).max(), np.ptp(out["logits_fp"])))


def gen_decode_case_w8pc_mha():
    """configs[2]-style recipe on the same graph: 8-bit PER-CHANNEL weights everywhere, and full multi-head attention (4 / 4 heads)."""
    gen_decode_case_w4(tag="w8pc_mha", wbits=8, kv_heads=4)


def gen_decode_case_gelu():
    """Gemma-style gated MLP (GeGLU: act_fn = GELU -> QGELU, qmodule.py:756-798, :856) with multi-query attention (4 / 1 heads), W4A8."""
    gen_decode_case_w4(tag="w4_geglu_mqa", wbits=4, kv_heads=1, act="gelu")


def gen_train_step():
    """f3: ONE inner step of e2equant (mobilellm/quantization/algorithm.py:727-745, the deployment recipe's flags: --lwc --let --lrl
    --deactive_amp, fp32, 4-bit per-channel weights: experiments/w4a8/main/e2e_llama-s1024-ep60.sh:17-23) on one decoder layer of the
    reference's HFForCausalLM: LET scales registered (:690-706), LWC enabled on every weight quantizer (enable_quant :325-351), scale /
    offset as nn.Parameters (LRL), smooth_lm_temporary, quantized forward, MSE against the layer's own fp output, backward -> the loss
    and the gradient of every trainable tensor (let / lwc / lrl groups of :239-282)."""
    import mobilellm.quantization.algorithm as A
    m, cfg = _tiny_hf(4, 91)
    g = torch.Generator().manual_seed(15)
    layer = m.model.layers[0]
    with torch.no_grad():
        for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj, layer.mlp.w1,
                    layer.mlp.w2, layer.mlp.w3):            # LET on q / k needs biases (algorithm.py:89-96)
            lin.bias = nn.Parameter(torch.randn(lin.out_features, generator=g) * 0.1)
    out = {f"sd|{k_}": npf(v_) for k_, v_ in layer.state_dict().items() if "rotary" not in k_}
    S = 24
    x = torch.randn(1, S, 64, generator=g)
    mask = torch.full((S, S), float("-inf")).triu(1)[None, None]
    pos = torch.arange(S)[None]
    with torch.no_grad():
        y_fp = layer(x, attention_mask=mask, position_ids=pos)[0]
    # ranges of every leaf the surgery will wrap, from this one forward (what generate_act_range.py's hooks record)
    act = {}
    def hook(name):
        def fn(mod, xx, yy):
            d = act.setdefault(name, {})
            d["input"] = [float(xx[0].min()), float(xx[0].max())]
            d["output"] = [float(yy.min()), float(yy.max())]
            if isinstance(mod, FMatMul):
                d["input2"] = [float(xx[1].min()), float(xx[1].max())]
        return fn
    from mobilellm.model.hf_model import HFRMSNorm
    hs = [mod.register_forward_hook(hook(n)) for n, mod in layer.named_modules() if isinstance(mod, (nn.Linear, nn.SiLU, HFRMSNorm, FMatMul))]
    with torch.no_grad():
        layer(x, attention_mask=mask, position_ids=pos)
    for h in hs:
        h.remove()
    Q.create_sim_qmodel(layer, Q.QuantConfig(bitwidth=4, is_per_channel=True), Q.QuantConfig(bitwidth=8))
    for name, mod in layer.named_modules():          # ptq/mobilequant.py:175-201
        if isinstance(mod, Q.QLinear):
            if any(k_ in name for k_ in ("q_proj", "k_proj", "v_proj", "o_proj", "w1", "w3")):
                mod.input_quantizer = None
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QSiLU):
            mod.input_quantizer = None
    Q.set_scale_and_offset(layer, act, "parameter")
    for name, mod in layer.named_modules():          # enable_quant(args with lwc): algorithm.py:325-351
        if isinstance(mod, (Q.QLinear, Q.QRMSNorm)):
            mod.weight_quantizer.enable_lwc(mod.weight)
    params = {}
    for name, dim in (("qkv", 64), ("fc1", 64), ("out", 64), ("fc2", 96), ("qkt", 64)):
        sc = torch.rand(dim, generator=g) * 0.6 + 0.7
        layer.register_parameter(f"{name}_smooth_scale", nn.Parameter(sc))
        params[f"{name}_smooth_scale"] = sc
        if name != "qkt":                                   # registered like the reference does, unused without --use_shift
            layer.register_parameter(f"{name}_smooth_shift", nn.Parameter(torch.zeros(dim)))
    for p_ in layer.parameters():
        p_.requires_grad_(True)
    with torch.enable_grad():
        A.smooth_lm_temporary(layer, cfg, True, False)
        y_q = layer(x, attention_mask=mask, position_ids=pos)[0]
        loss = torch.nn.MSELoss()(y_fp, y_q)
        loss.backward()
    grads = {}
    for name, p_ in layer.named_parameters():
        if any(t in name for t in ("bound_factor", "smooth_scale", "quantizer.offset", "quantizer.scale")):      # get_parameters :264-271
            assert p_.grad is not None, name
            grads[name] = p_.grad
    for k_, v_ in params.items():
        out["let|" + k_] = npf(v_)
    for k_, v_ in grads.items():
        out["grad|" + k_] = npf(v_)
    out.update(x=npf(x), y_fp=npf(y_fp), y_q=npf(y_q.detach()), loss=np.float64(loss.item()), act=np.array(json.dumps(act)))
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("train_step: loss %.6f, %d gradient tensors, groups:" % (loss.item(), len(grads)),
          {t: sum(t in k_ for k_ in grads) for t in ("bound_factor", "smooth_scale", "quantizer.scale", "quantizer.offset")})


def gen_generate_case():
    """f2: SimModel.generate's loop (mobilellm/model/sim_model.py:160-221: next token = argmax of the logits behind the context, append,
    stop at EOS, feed the token back) run FREE on the reference's W8A8-simulated HFForCausalLM of decode_case.npz (same weights,
    ranges and qcfg, loaded from that fixture): the greedy continuation of an 8-token context, 12 new tokens, each step a full
    forward of the reference model (no cache: the same function of the tokens so far).  Of 300 random contexts the one whose
    smallest top-1 / top-2 margin is largest is kept (> 1 % of the logit span: the decode tests hold the logits to a median of 0.2 %),
    so a faithful implementation does not flip a token on rounding noise."""
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    z = np.load(os.path.join(OUT, "decode_case.npz"), allow_pickle=False)
    cfg = HFConfig(vocab_size=96, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=2, max_position_embeddings=64, hidden_act="silu", use_matmul_as_module=True)
    cfg._attn_implementation = "eager"
    m = HFForCausalLM(cfg).eval()
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd|")}
    m.load_state_dict(sd, strict=True)
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8))
    Q.update_qcfg(m, json.loads(str(z["qcfg"])))
    Q.set_scale_and_offset(m, json.loads(str(z["act"])), "buffer")
    span = float(np.ptp(z["logits_fp"]))
    g = torch.Generator().manual_seed(77)
    best = None
    for attempt in range(300):
        ctx = torch.randint(3, 96, (8,), generator=g)
        toks, margins = ctx.tolist(), []
        for _ in range(12):
            logits = m(torch.tensor(toks)[None], use_cache=False).logits[0, -1]
            top = torch.topk(logits, 2).values
            margins.append(float(top[0] - top[1]) / span)
            toks.append(int(torch.argmax(logits)))          # sim_model.py:200-201
            if margins[-1] < (best[0] if best else 0.0):
                break
        if len(margins) == 12 and (best is None or min(margins) > best[0]):
            best = (min(margins), toks, margins)
    assert best is not None and best[0] > 0.01, best
    _, toks, margins = best
    np.savez_compressed(os.path.join(OUT, "generate_case.npz"), context=np.array(toks[:8], np.int64), tokens=np.array(toks, np.int64),
                        margins=np.array(margins, np.float32))
    print("generate_case: tokens", toks, "min margin %.4f of span" % min(margins))


def gen_decode_case_stablelm():
    """BASELINE.json configs[2], the StableLM-2 leaf graph (hf_config.py: norm_class = layernorm, attention_bias + use_qkv_bias_only,
    partial_rotary_factor = 0.25) at toy size: nn.LayerNorm -> QLayerNorm (qmodule.py:861-862), biased q / k / v, RoPE on the first
    16 of 64 head dims, full multi-head attention; 8-bit PER-CHANNEL weights, the mixed-precision rules of ptq/mobilequant.py:175-201."""
    gen_decode_case_w4(tag="stablelm", wbits=8, kv_heads=4,
                       cfg_kw=dict(norm_class="layernorm", attention_bias=True, use_qkv_bias_only=True, partial_rotary_factor=0.25))


def gen_decode_case_gemma():
    """BASELINE.json configs[3], the Gemma leaf graph (sim_model.py:45-46 at toy size): explicit head_dim 256 (2 heads / 1 KV head:
    heads * head_dim != hidden), GeGLU, embeddings scaled by sqrt(hidden) (normalize_embed); 4-bit per-channel SYMMETRIC weights
    (experiments/w4a8/main/e2e_gemma-s1024-ep60-sym.sh:23), 8-bit activations."""
    gen_decode_case_w4(tag="gemma", wbits=4, kv_heads=1, act="gelu", heads=2, wsym=True, cfg_kw=dict(head_dim=256, normalize_embed=True))


def gen_layer_case():
    """The whole simulated-quant forward at BASELINE size: ONE TinyLlama-1.1B decoder layer (hidden 2048, 32 heads / 4 KV heads,
    head_dim 64, FFN 5632 -- mobilellm/model/sim_model.py:43-44) of the REAL reference HFForCausalLM on a 2048-token sequence, W8A8
    recipe of ptq/mobilequant.py:175-201, ranges from the reference's own get_act_range.  Weights come from
    tests/toy_models.seeded_parameters_ (regenerated on the test side), so the fixture holds only ids, ranges, configs and the
    logits of every 8th position over a 128-word vocabulary."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from seeded import seeded_parameters_
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    S, V = 2048, 128
    cfg = HFConfig(vocab_size=V, hidden_size=2048, intermediate_size=5632, num_hidden_layers=1, num_attention_heads=32,
                   num_key_value_heads=4, max_position_embeddings=S, hidden_act="silu", use_matmul_as_module=True)
    cfg._attn_implementation = "eager"
    m = HFForCausalLM(cfg).eval()
    seeded_parameters_(m, std=0.05, strip="model.")
    g = torch.Generator().manual_seed(31)
    ids = torch.randint(0, V, (1, S), generator=g)
    calib = [torch.randint(0, V, (1, S), generator=g), ids]
    out = {"ids": npf(ids[0])}
    with torch.no_grad():
        out["logits_fp"] = npf(m(ids, use_cache=False).logits[0, ::8])
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _Tk:
        bos_token_id, vocab_size = 1, V
        def __call__(s_, line, return_tensors="pt", max_length=None, truncation=True):
            return types.SimpleNamespace(input_ids=calib[int(line)])
    _orig = m.forward
    m.forward = lambda x_, **kw: _orig(x_, use_cache=False)
    act = rng_mod.get_act_range(m, _Tk(), [{"text": str(i)} for i in range(len(calib))], len(calib), S)
    m.forward = _orig
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8))
    for name, mod in m.named_modules():          # ptq/mobilequant.py:175-201
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    act = {k_: v_ for k_, v_ in act.items() if any(k_ == n for n, mm in m.named_modules() if isinstance(mm, (Q.QLinear, Q.QRMSNorm, Q.QMatMul, Q.QSiLU)))}
    Q.set_scale_and_offset(m, act, "buffer")
    with torch.no_grad():
        out["logits_w8a8"] = npf(m(ids, use_cache=False).logits[0, ::8])
    out["act"] = np.array(json.dumps(act))
    out["qcfg"] = np.array(json.dumps(Q.export_qcfg(m)))
    np.savez_compressed(os.path.join(OUT, "layer_case.npz"), **out)
    d = out["logits_w8a8"] - out["logits_fp"]
    print("layer_case: logits", out["logits_w8a8"].shape, "max |w8a8 - fp| %.4f of span %.3f" % (np.abs(d).max(), np.ptp(out["logits_fp"])))


def gen_full_depth_case(S=256, V=512, layers=22):
    """Model-DEPTH parity ("quantized perplexity within 0.05 of reference" is a full-model bar, eval/harness_eval.py:
    75-108): the REAL reference HFForCausalLM at TinyLlama-1.1B's geometry -- 22 layers, hidden 2048, 32 heads / 4 KV heads, head_dim 64,
    FFN 5632 (mobilellm/model/sim_model.py:43-44) -- with a reduced vocabulary, on a 256-token sequence; ranges from the reference's own
    get_act_range; then BOTH deployment recipes on the same calibrated model: W8A8 (per-tensor weights, ptq/mobilequant.py:175-201 mixed
    precision) and W4A8 (4-bit per-channel asymmetric weights, experiments/w4a8/main/e2e_llama-s1024-ep60.sh:23).  Weights come from
    tests/seeded.py (regenerated on the test side).  Stored: ids, ranges, qcfg, and per recipe the next-token NLL of every position
    (float64), the argmax of every position and the logits of every 8th position over the whole vocabulary -- how index flips accumulate
    over 22 layers is what this fixture pins -- and the same quantities of a SECOND reference run with three BLAS threads (`*_self3`): the
    reference's own reproducibility floor, which is what an implementation can be held to."""
    _full_depth("full_depth_case", S, V, layers, stable=False)


def gen_full_depth_stable_case(S=256, V=512, layers=22):
    """The 0.05-perplexity bar of BASELINE.json made TESTABLE at depth.  full_depth_case.npz showed that on a
    random-weight 22-layer model the reference does not reproduce itself (chaotic: each branch has O(1) gain on a residual stream it
    dominates, the logits are flat).  A trained checkpoint is contractive where that model is not -- the residual stream carries the
    token identity, every branch adds a small correction, the unembedding is peaked -- and no checkpoint is reachable offline, so
    this fixture BUILDS such a model at TinyLlama-1.1B's geometry (tests/seeded.py: seeded_contractive_parameters_): unit-variance
    embeddings, o_proj / w2 scaled so a branch adds ~0.25 of the stream's RMS, row-wise decaying projections, and an unembedding
    that reads the embedding of the PREDECESSOR under a fixed vocabulary permutation (a bigram model: the token after t is pi(t)).
    The evaluated sequence follows pi with probability 0.7 and jumps otherwise, so the fp perplexity sits where a small LM's does
    (fp 12.7) and responds to every logit.  Same reference calls as full_depth_case (get_act_range, create_sim_qmodel, the
    mixed-precision rules of ptq/mobilequant.py:175-201, W8A8 and W4A8), same stored quantities, and again the reference's own second
    run with three BLAS threads: here the two runs agree to < 0.01 perplexity, so `|delta ppl| <= 0.05` is a HARD bar for the module
    chain, the fused prefill and the decode engine (tests/test_gpu_round5.py), with no self-calibrated yardstick."""
    _full_depth("full_depth_stable_case", S, V, layers, stable=True)


def _full_depth(fname, S, V, layers, stable):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import copy
    import time as _t
    from seeded import seeded_parameters_, seeded_contractive_parameters_, bigram_sequence
    from mobilellm.model.hf_config import HFConfig
    from mobilellm.model.hf_model import HFForCausalLM
    cfg = HFConfig(vocab_size=V, hidden_size=2048, intermediate_size=5632, num_hidden_layers=layers, num_attention_heads=32,
                   num_key_value_heads=4, max_position_embeddings=S, hidden_act="silu", use_matmul_as_module=True)
    cfg._attn_implementation = "eager"
    m = HFForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(77)
    if stable:
        seeded_contractive_parameters_(m, strip="model.")
        ids = bigram_sequence(V, S, g)
        calib = [bigram_sequence(V, S, g), bigram_sequence(V, S, g), ids]
    else:
        seeded_parameters_(m, std=0.02, strip="model.")
        ids = torch.randint(0, V, (1, S), generator=g)
        calib = [torch.randint(0, V, (1, S), generator=g), torch.randint(0, V, (1, S), generator=g), ids]
    out = {"ids": npf(ids[0])}
    # the stable case evaluates SEVEN more sequences (drawn after the calibration set, which stays as it was): 2 040 predicted tokens
    # instead of 255 -- the perplexity of 255 positions carries ~0.04 of position-correlated summation-order noise (an index flip in a
    # cached key / value moves every later position), a hard 0.05 bar needs the average over more text
    more = torch.cat([bigram_sequence(V, S, g) for _ in range(7)]) if stable else None
    if more is not None:
        out["ids_more"] = npf(more)

    def stats(run, tag):
        logits = run(ids)
        lg = logits[0].double()
        nll = -(torch.log_softmax(lg[:-1], -1).gather(1, ids[0, 1:, None])[:, 0])
        out["nll_" + tag] = nll.numpy()
        out["argmax_" + tag] = lg.argmax(-1).numpy().astype(np.int32)
        out["logits_" + tag] = npf(logits[0, ::8])
        print(f"{fname}[{tag}]: NLL {float(nll.mean()):.6f}  ppl {float(nll.mean().exp()):.4f}", flush=True)
        if more is not None:
            nlls, args = [], []
            for i in range(more.shape[0]):
                lgi = run(more[i:i + 1])[0].double()
                nlls.append(-(torch.log_softmax(lgi[:-1], -1).gather(1, more[i, 1:, None])[:, 0]).numpy())
                args.append(lgi.argmax(-1).numpy().astype(np.int32))
            out["nll_more_" + tag], out["argmax_more_" + tag] = np.stack(nlls), np.stack(args)
            allnll = np.concatenate([nll.numpy()] + nlls)
            print(f"{fname}[{tag}]: all {allnll.size} positions: NLL {allnll.mean():.6f}  ppl {np.exp(allnll.mean()):.4f}", flush=True)
    t0 = _t.time()
    torch.set_num_threads(1)
    with torch.no_grad():
        stats(lambda x_: m(x_, use_cache=False).logits, "fp")
    rng_mod = _load_script(os.path.join(REF, "ptq", "generate_act_range.py"), ["x", "--hf_path", "none"])
    rng_mod.args.per_channel = False

    class _Tk:
        bos_token_id, vocab_size = 1, V
        def __call__(s_, line, return_tensors="pt", max_length=None, truncation=True):
            return types.SimpleNamespace(input_ids=calib[int(line)])
    _orig = m.forward
    m.forward = lambda x_, **kw: _orig(x_, use_cache=False)
    act = rng_mod.get_act_range(m, _Tk(), [{"text": str(i)} for i in range(len(calib))], len(calib), S)
    m.forward = _orig
    print(f"{fname}: fp forward + calibration {_t.time() - t0:.0f} s", flush=True)
    for tag, wcfg in (("w8a8", Q.QuantConfig(bitwidth=8)), ("w4a8", Q.QuantConfig(bitwidth=4, is_per_channel=True))):
        mq_ = copy.deepcopy(m)
        Q.create_sim_qmodel(mq_, wcfg, Q.QuantConfig(bitwidth=8))
        for name, mod in mq_.named_modules():          # ptq/mobilequant.py:175-201
            if isinstance(mod, Q.QLinear):
                if "w2" in name:
                    mod.weight_quantizer.qcfg.is_per_channel = True
                    mod.output_quantizer.qcfg.bitwidth = 16
                elif "o_proj" in name:
                    mod.output_quantizer.qcfg.bitwidth = 16
            elif isinstance(mod, Q.QRMSNorm):
                mod.input_quantizer.qcfg.bitwidth = 16
                mod.weight_quantizer.qcfg.bitwidth = 16
                mod.weight_quantizer.qcfg.is_symmetric = False
                mod.weight_quantizer.qcfg.is_per_channel = False
            elif isinstance(mod, Q.QMatMul):
                if "qk_bmm" in name:
                    mod.output_quantizer.qcfg.bitwidth = 16
                if "pv_bmm" in name:
                    mod.input_quantizer.qcfg.bitwidth = 16
        a_ = {k_: v_ for k_, v_ in act.items() if any(k_ == n for n, mm in mq_.named_modules()
                                                      if isinstance(mm, (Q.QLinear, Q.QRMSNorm, Q.QMatMul, Q.QSiLU)))}
        Q.set_scale_and_offset(mq_, a_, "buffer")
        t0 = _t.time()
        prev = torch.get_num_threads()
        with torch.no_grad():
            torch.set_num_threads(1)                 # the canonical run: one thread = one summation order
            stats(lambda x_: mq_(x_, use_cache=False).logits, tag)
            # the reference against ITSELF: the same forward with the BLAS splitting its dot products differently.  Fake-quant behind
            # fp32 matmuls is not reproducible to the index (K = 2048 ... 5632 sums carry ~1e-5 of the output scale, an 8-bit LSB is
            # 4e-3: ~0.25 % of all 8-bit indices flip), and 22 layers amplify that: this spread is the floor any implementation sits on
            torch.set_num_threads(3)
            stats(lambda x_: mq_(x_, use_cache=False).logits, tag + "_self3")
            torch.set_num_threads(prev)
        print(f"{fname}[{tag}]: simulated forwards {_t.time() - t0:.0f} s", flush=True)
        out["qcfg_" + tag] = np.array(json.dumps(Q.export_qcfg(mq_)))
        out["act"] = np.array(json.dumps(a_))
        del mq_
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), **out)
    print(fname + ": %.0f KiB" % (os.path.getsize(os.path.join(OUT, fname + ".npz")) / 1024))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:                                  # python oracle/gen_golden.py gen_lwc_cases ...: regenerate selected fixtures
        for name in only:
            globals()[name]()
        sys.exit(0)
    gen_lwc_cases()
    gen_qmatmul_cases()
    gen_toy_lm_nll()
    gen_smooth_cases()
    gen_decode_case()
    gen_decode_case_w4()
    gen_decode_case_w8pc_mha()
    gen_decode_case_gelu()
    gen_generate_case()
    gen_train_step()
    gen_decode_case_stablelm()
    gen_decode_case_gemma()
    gen_layer_case()
    gen_full_depth_case()
    gen_full_depth_stable_case()
    gen_scale_offset_grid()
    gen_quantizer_cases()
    gen_nonfinite()
    gen_qrmsnorm_cases()
    gen_qact_cases()
    gen_quantizer_grads()
    gen_qlinear_cases()
    gen_qlinear_dynamic_cases()
    gen_qlinear_grouped_cases()
    gen_calib_stream()
    gen_checksums()
    gen_api_surface()
    gen_artifacts()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", os.path.normpath(OUT), f"({tot/1024:.0f} KiB)")
