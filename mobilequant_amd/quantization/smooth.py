"""Learnable equivalent transformation (LET) plumbing on the device: MI355X-side mirror of
``mobilellm/quantization/algorithm.py:47-233``.

The reference's PTQ loops (``omniquant`` / ``e2equant``, algorithm.py:381 / :587) call ``smooth_lm_temporary`` before every
forward: the per-channel LET scales / shifts registered on a decoder layer (``qkv_smooth_scale`` ...; algorithm.py:692-706)
are applied to the float weights and parked as ``temp_weight`` / ``temp_bias`` on the Q-modules, which then run with
``use_temporary_parameter = True``; ``smooth_lm_inplace`` folds them for good and clamps the weights to their learned clipping
range (``run_lwc``).  Same function names, arguments and attribute names here, operating on this package's Q-modules; the
algebra is weight-sized elementwise torch ops on device tensors and keeps autograd (the scales are what is being trained).
The Q-modules pick the temporary parameters up in their forward (``QLinear`` / ``QRMSNorm`` / ``QLayerNorm``), the integer
weight cache re-quantises when ``temp_weight`` changes.
"""
from __future__ import annotations

import torch

from .qmodule import QLayerNorm, QLinear, QRMSNorm

__all__ = ["truncate_number", "smooth_ln_fcs_temporary", "smooth_fc_fc_temporary", "smooth_q_k_temporary",
           "smooth_ln_fcs_inplace", "smooth_fc_fc_inplace", "smooth_q_k_inplace", "smooth_lm_temporary", "smooth_lm_inplace"]


class _Truncate(torch.autograd.Function):
    """|x| < threshold -> sign(x) * threshold, identity gradient (algorithm.py:26-42; keeps AMP training from overflowing)."""

    @staticmethod
    def forward(ctx, x, threshold):
        # the reference writes out[small] = out[small].sign() * threshold (boolean-mask indexing: a nonzero() and a host sync per
        # parameter and step); the same values as one select, no sync
        return torch.where(x.abs() < threshold, x.sign() * threshold, x)

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


def truncate_number(number, threshold: float = 1e-2):
    return _Truncate.apply(number, threshold)


def _bias_or_zero(m):
    b = getattr(m, "bias", None)
    return b if b is not None else 0


def _as_list(fcs):
    return list(fcs) if isinstance(fcs, (list, tuple)) else [fcs]


# ---- norm -> linears -------------------------------------------------------------------------------------------------------
def smooth_ln_fcs_temporary(ln, fcs, scales, shifts):
    """``y = (x - shift) / scale`` absorbed by the norm, undone by the linears (algorithm.py:47-68)."""
    ln.use_temporary_parameter = True
    ln.temp_bias = (_bias_or_zero(ln) - shifts) / scales if getattr(ln, "bias", None) is not None else (-1 * shifts) / scales
    ln.temp_weight = ln.weight / scales
    for fc in _as_list(fcs):
        fc.use_temporary_parameter = True
        fc.temp_bias = (fc.bias + fc.weight @ shifts) if fc.bias is not None else fc.weight @ shifts
        fc.temp_weight = fc.weight * scales.view(1, -1)


@torch.no_grad()
def smooth_ln_fcs_inplace(ln, fcs, scales, shifts):
    """algorithm.py:100-122."""
    ln.use_temporary_parameter = False
    if getattr(ln, "bias", None) is not None:
        ln.bias.sub_(shifts)
        ln.bias.div_(scales)
    else:
        if hasattr(ln, "bias"):
            del ln.bias
        ln.register_buffer("bias", (-1 * shifts) / scales)
    ln.weight.div_(scales)
    for fc in _as_list(fcs):
        fc.use_temporary_parameter = False
        if fc.bias is not None:
            fc.bias.add_(fc.weight @ shifts)
        else:
            del fc.bias
            fc.register_buffer("bias", fc.weight @ shifts)
        fc.weight.mul_(scales.view(1, -1))


# ---- linear -> linear (v_proj -> o_proj, w3 -> w2) ---------------------------------------------------------------------------
def smooth_fc_fc_temporary(fc1, fc2, scales, shifts):
    """algorithm.py:71-87 (fc1 may already carry temporaries from the norm pair in front of it)."""
    fc1.use_temporary_parameter = True
    fc2.use_temporary_parameter = True
    if hasattr(fc1, "temp_weight"):
        fc1.temp_bias = (fc1.temp_bias - shifts) / scales.view(-1)
        fc1.temp_weight = fc1.temp_weight / scales.view(-1, 1)
    else:
        fc1.temp_bias = (fc1.bias - shifts) / scales.view(-1)
        fc1.temp_weight = fc1.weight / scales.view(-1, 1)
    fc2.temp_bias = (fc2.bias + fc2.weight @ shifts) if fc2.bias is not None else fc2.weight @ shifts
    fc2.temp_weight = fc2.weight * scales.view(1, -1)


@torch.no_grad()
def smooth_fc_fc_inplace(fc1, fc2, scales, shifts):
    """algorithm.py:125-137."""
    fc1.use_temporary_parameter = False
    fc2.use_temporary_parameter = False
    fc1.bias.sub_(shifts)
    fc1.bias.div_(scales.view(-1))
    fc1.weight.div_(scales.view(-1, 1))
    if fc2.bias is not None:
        fc2.bias.add_(fc2.weight @ shifts)
    else:
        del fc2.bias
        fc2.register_buffer("bias", fc2.weight @ shifts)
    fc2.weight.mul_(scales.view(1, -1))


# ---- q <-> k ---------------------------------------------------------------------------------------------------------------
def smooth_q_k_temporary(q_proj, k_proj, scales):
    """algorithm.py:90-97."""
    q_proj.use_temporary_parameter = True
    k_proj.use_temporary_parameter = True
    q_proj.temp_weight = q_proj.temp_weight / scales.view(-1, 1)
    q_proj.temp_bias = q_proj.temp_bias / scales.view(-1)
    k_proj.temp_weight = k_proj.temp_weight * scales.view(-1, 1)
    k_proj.temp_bias = k_proj.temp_bias * scales.view(-1)


@torch.no_grad()
def smooth_q_k_inplace(q_proj, k_proj, scales):
    """algorithm.py:140-146."""
    q_proj.use_temporary_parameter = False
    k_proj.use_temporary_parameter = False
    q_proj.weight.div_(scales.view(-1, 1))
    q_proj.bias.div_(scales.view(-1))
    k_proj.weight.mul_(scales.view(-1, 1))
    k_proj.bias.mul_(scales.view(-1))


# ---- one decoder layer -------------------------------------------------------------------------------------------------------
def _pairs(layer, config):
    """(kind, producer, consumers, scale name, shift name) in the reference's order (algorithm.py:155-179 / :196-220)."""
    attn, mlp = layer.self_attn, layer.mlp
    three = getattr(config, "num_linears_per_mlp", 3 if hasattr(mlp, "w3") else 2) == 3
    ffn_in = [mlp.w1] + ([mlp.w3] if three else [])
    qkv = [attn.q_proj, attn.k_proj, attn.v_proj]
    if getattr(config, "shared_attention_norm", False):
        yield "ln", layer.input_layernorm, qkv + ffn_in, "qkv"
    else:
        yield "ln", layer.input_layernorm, qkv, "qkv"
        yield "ln", layer.post_attention_layernorm, ffn_in, "fc1"
    if attn.v_proj.weight.shape[0] == attn.o_proj.weight.shape[1]:
        yield "fc", attn.v_proj, attn.o_proj, "out"
    return three


def _truncate_smooth_params(layer, use_shift):
    template = "smooth" if use_shift else "smooth_scale"
    for name, p in layer.named_parameters():
        if template in name:
            p.data = truncate_number(p)


def smooth_lm_temporary(model, config, use_let, use_shift=False, original_omniquant=False):
    """Park the LET-transformed weights on the Q-modules of one decoder layer (algorithm.py:187-233).  ``model`` is the layer
    carrying ``qkv_smooth_scale`` / ``qkv_smooth_shift`` / ``fc1_...`` / ``out_...`` / ``fc2_...`` / ``qkt_smooth_scale``."""
    if use_let:
        with torch.no_grad():
            _truncate_smooth_params(model, use_shift)
        attn, mlp = model.self_attn, model.mlp
        for kind, prod, cons, key in _pairs(model, config):
            sc, sh = getattr(model, key + "_smooth_scale"), getattr(model, key + "_smooth_shift")
            (smooth_ln_fcs_temporary if kind == "ln" else smooth_fc_fc_temporary)(prod, cons, sc, sh)
        three = getattr(config, "num_linears_per_mlp", 3 if hasattr(mlp, "w3") else 2) == 3
        if three and not original_omniquant:
            smooth_fc_fc_temporary(mlp.w3, mlp.w2, model.fc2_smooth_scale, model.fc2_smooth_shift)
        if attn.q_proj.weight.shape[0] == attn.k_proj.weight.shape[0]:
            smooth_q_k_temporary(attn.q_proj, attn.k_proj, model.qkt_smooth_scale)
    else:
        for _, m in model.named_modules():
            if isinstance(m, QLinear):
                m.temp_weight, m.temp_bias = m.weight, m.bias
    for _, m in model.named_modules():
        if isinstance(m, QLinear):
            m.use_temporary_parameter = True
            if not hasattr(m, "temp_bias"):
                m.temp_bias = m.bias
            if not hasattr(m, "temp_weight"):
                m.temp_weight = m.weight


@torch.no_grad()
def smooth_lm_inplace(model, config, use_let, use_shift=False, original_omniquant=False):
    """Fold the LET parameters for good and clamp every weight to its learned clipping range (algorithm.py:147-184)."""
    if use_let:
        _truncate_smooth_params(model, use_shift)
        attn, mlp = model.self_attn, model.mlp
        for kind, prod, cons, key in _pairs(model, config):
            sc, sh = getattr(model, key + "_smooth_scale"), getattr(model, key + "_smooth_shift")
            (smooth_ln_fcs_inplace if kind == "ln" else smooth_fc_fc_inplace)(prod, cons, sc, sh)
        three = getattr(config, "num_linears_per_mlp", 3 if hasattr(mlp, "w3") else 2) == 3
        if three and not original_omniquant:
            smooth_fc_fc_inplace(mlp.w3, mlp.w2, model.fc2_smooth_scale, model.fc2_smooth_shift)
        if attn.q_proj.weight.shape[0] == attn.k_proj.weight.shape[0]:
            smooth_q_k_inplace(attn.q_proj, attn.k_proj, model.qkt_smooth_scale)
    for _, m in model.named_modules():
        if isinstance(m, (QLinear, QRMSNorm, QLayerNorm)):
            m.weight.data = m.weight_quantizer.run_lwc(m.weight)
            m.use_temporary_parameter = False
