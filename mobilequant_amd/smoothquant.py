"""SmoothQuant on the device: activation statistics -> per-channel scales -> fold into the weights, and the artefacts the
reference's scripts exchange.  MI355X-native counterpart of ``ptq/generate_act_scale_shift.py:42-93`` (absmax statistics) and
``ptq/smoothquant.py:50-139`` (the fold), SURVEY section 8f rank 4.

  * statistics: ``get_act_scales`` rides on the calibration collector (``mobilequant_amd.calibration``): one single-pass HIP
    column reduction per hooked tensor into device-resident running statistics, samples sharded over ranks, ONE all-reduce;
    the reference copies every hooked tensor's absmax to the CPU in every hook (generate_act_scale_shift.py:49-53).
  * fold: ``s = act_absmax^alpha / weight_absmax^(1-alpha)`` with the weight absmax from the HIP column reduction, then
    ``norm.weight /= s`` (or ``fc1.weight /= s`` per output row) and ``fc.weight *= s`` per input column, in place on the
    device (smoothquant.py:50-105).  One-shot weight algebra: torch elementwise ops on device tensors.
  * artefacts: ``act_scales.pth`` (dict ``"<module>_<input|output>" -> Tensor[C]`` on the CPU, generate_act_scale_shift.py:49-53,
    :170-175), ``act_dict_per_channel.pth`` (``{module: {field: Tensor[2, C]}}``, generate_act_range.py:155-158), ``act_dict.json``
    (``json_save``: indent 4, sorted keys, mobilellm/utils/io.py:34-36).

The scales are folded into weights OFFLINE in the reference: there is no run-time per-channel activation multiply on its path.
For activations whose producer cannot absorb ``1/s`` this package also offers the run-time form -- ``x / s`` fused into the
activation quantize kernel (``ops.quantize(..., chan_scale=s)``, ``QLinear.set_input_channel_scale``).
"""
from __future__ import annotations

import json
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from .calibration import ActRangeCollector, _nullcontext, save_act_dict  # noqa: F401  (save_act_dict: json_save's bytes)
from .quantization.fp_ops import HFRMSNorm

__all__ = ["get_act_scales", "smooth_ln_fcs", "smooth_fc_fcs", "smooth_lm", "save_act_scales", "save_act_dict_per_channel",
           "save_act_dict", "load_act_scales"]


# ---- statistics --------------------------------------------------------------------------------------------------------
@torch.no_grad()
def get_act_scales(model: nn.Module, samples: Sequence[torch.Tensor], group=None, forward=None) -> Dict[str, torch.Tensor]:
    """Per-channel absmax of the input / output of every Linear / LayerNorm / RMSNorm leaf over ``samples`` (identical list on
    every rank; rank r runs samples r, r + world, ...), keyed ``"<module>_<field>"`` like the reference's ``act_scales.pth``."""
    import torch.distributed as dist
    model.eval()
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
    col = ActRangeCollector(model, per_channel=True).attach()
    dev = col.device
    run = forward if forward is not None else (lambda s: model(s))
    mine = list(range(rank, len(samples), world)) or ([rank % len(samples)] if len(samples) else [])
    try:
        for i in mine:
            with torch.cuda.device(dev) if dev.type == "cuda" else _nullcontext():
                run(samples[i].to(dev))
    finally:
        col.detach()
    col.all_reduce(group)
    return col.act_scales()


@torch.no_grad()
def get_act_shifts(model: nn.Module, samples: Sequence[torch.Tensor], forward=None) -> Dict[str, torch.Tensor]:
    """``get_act_shifts`` (ptq/generate_act_scale_shift.py:97-149): per Linear / LayerNorm / RMSNorm leaf and per channel, the running
    average ``0.99 * shift + 0.01 * (max + min) / 2`` of each sample's per-channel mid-range (the first sample initialises it), for
    the input and the output, keyed ``"<module>_<field>"`` -- what ``--use_shift`` LET reads from ``act_shifts.pth``.  The average
    depends on the sample ORDER, so unlike the min / max statistics it is not sharded: one stream, in order.  Per hook the reference
    takes two reductions and two device-to-host copies; here one pass of the HIP column reduction leaves [min, max] per channel on
    the device, the update is three fp32 elementwise ops there (the reference's, in its order: bit-identical), and the host sees
    the result once."""
    model.eval()
    dev = next(model.parameters()).device
    col = ActShiftCollector()

    def hook(name):
        def fn(m, xx, yy):
            col.update(name, "input", xx[0] if isinstance(xx, tuple) else xx)
            col.update(name, "output", yy[0] if isinstance(yy, tuple) else yy)
        return fn
    hooks = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if isinstance(m, nn.Linear) or _is_norm(m)]
    run = forward if forward is not None else (lambda s_: model(s_))
    try:
        for s_ in samples:
            with torch.cuda.device(dev) if dev.type == "cuda" else _nullcontext():
                run(s_.to(dev))
    finally:
        for h in hooks:
            h.remove()
    return col.result()


class ActShiftCollector:
    """The running statistic of get_act_shifts, one tensor at a time (generate_act_scale_shift.py:102-111)."""

    def __init__(self):
        self.shifts: Dict[str, torch.Tensor] = {}

    def update(self, name: str, field: str, t: torch.Tensor) -> None:
        t2 = t.detach().reshape(-1, t.shape[-1])
        if t2.dtype != torch.float32 or not t2.is_contiguous():
            t2 = t2.float().contiguous()
        mn, mx = ops.minmax_cols(t2)                       # one pass: [min, max] per channel, on the device
        new = (mx + mn) / 2
        key = f"{name}_{field}"
        self.shifts[key] = 0.99 * self.shifts[key] + 0.01 * new if key in self.shifts else new

    def result(self) -> Dict[str, torch.Tensor]:
        return {k: v.float().cpu() for k, v in self.shifts.items()}


def save_act_shifts(path: str, act_shifts: Dict[str, torch.Tensor]) -> None:
    """``act_shifts.pth`` as generate_act_scale_shift.py:177-180 writes it (torch.save of the CPU fp32 dictionary)."""
    torch.save({k: v.detach().float().cpu() for k, v in act_shifts.items()}, path)


# ---- fold ----------------------------------------------------------------------------------------------------------------
def _is_norm(m) -> bool:
    return isinstance(m, (nn.LayerNorm, HFRMSNorm)) or any(c.__name__ == "HFRMSNorm" for c in type(m).__mro__)


def _column_absmax(fcs: Iterable[nn.Linear]) -> torch.Tensor:
    """max over the linears of the per-input-channel |weight| maximum (smoothquant.py:60-61), by the HIP column reduction."""
    best = None
    for fc in fcs:
        mn, mx = ops.minmax_cols(fc.weight.detach())
        cur = torch.maximum(mn.abs(), mx.abs())
        best = cur if best is None else torch.maximum(best, cur)
    return best.clamp(min=1e-5)


def _scales(act_scales: torch.Tensor, fcs: List[nn.Linear], alpha: float) -> torch.Tensor:
    w = fcs[0].weight
    act = act_scales.to(device=w.device, dtype=w.dtype)
    return (act.pow(alpha) / _column_absmax(fcs).to(w.dtype).pow(1 - alpha)).clamp(min=1e-5)


@torch.no_grad()
def smooth_ln_fcs(ln, fcs, act_scales, alpha: float = 0.5):
    """norm -> linears: ``ln.weight /= s`` (bias too), ``fc.weight *= s`` per input channel (smoothquant.py:50-76)."""
    fcs = fcs if isinstance(fcs, (list, tuple)) else [fcs]
    assert _is_norm(ln) and all(isinstance(fc, nn.Linear) for fc in fcs)
    assert all(len(ln.weight) == fc.in_features == len(act_scales) for fc in fcs)
    s = _scales(act_scales, list(fcs), alpha)
    ln.weight.div_(s)
    if getattr(ln, "bias", None) is not None:
        ln.bias.div_(s)
    for fc in fcs:
        fc.weight.mul_(s.view(1, -1))
    _assert_finite(ln, *fcs)
    return s


@torch.no_grad()
def smooth_fc_fcs(fc1, fcs, act_scales, alpha: float = 0.5):
    """linear -> linears (v_proj -> o_proj, w3 -> w2; not in the original SmoothQuant): ``fc1.weight /= s`` per output row
    (bias too), ``fc.weight *= s`` per input channel (smoothquant.py:82-105)."""
    fcs = fcs if isinstance(fcs, (list, tuple)) else [fcs]
    s = _scales(act_scales, list(fcs), alpha)
    fc1.weight.div_(s.view(-1, 1))
    if fc1.bias is not None:
        fc1.bias.div_(s.view(-1))
    for fc in fcs:
        fc.weight.mul_(s.view(1, -1))
    _assert_finite(fc1, *fcs)
    return s


def _assert_finite(*modules):
    for m in modules:
        for p in m.parameters():
            assert not torch.isnan(p).any(), "NaN after smoothing"


def _is_decoder_layer(m) -> bool:
    return all(hasattr(m, a) for a in ("input_layernorm", "self_attn", "mlp")) and hasattr(m.self_attn, "q_proj")


@torch.no_grad()
def smooth_lm(model, scales: Dict[str, torch.Tensor], alpha: float = 0.5, original_smoothquant: bool = False,
              original_omniquant: bool = False):
    """Fold SmoothQuant scales into every decoder layer of ``model`` (smoothquant.py:109-139).  ``scales``: the act_scales
    dictionary (``"<layer>.self_attn.q_proj_input"`` ...).  Decoder layers are recognised by their sub-module names
    (input_layernorm / self_attn / mlp), so the reference's HFDecoderLayer and this package's llama.DecoderLayer both qualify;
    ``shared_attention_norm`` / ``num_linears_per_mlp`` come from ``model.config`` when present."""
    cfg = getattr(model, "config", None)
    shared = bool(getattr(cfg, "shared_attention_norm", False))
    for name, layer in model.named_modules():
        if not _is_decoder_layer(layer):
            continue
        attn, mlp = layer.self_attn, layer.mlp
        three = hasattr(mlp, "w3") if cfg is None or not hasattr(cfg, "num_linears_per_mlp") else cfg.num_linears_per_mlp == 3
        qkv = [attn.q_proj, attn.k_proj, attn.v_proj]
        ffn_in = [mlp.w1] + ([mlp.w3] if three else [])
        if shared:
            smooth_ln_fcs(layer.input_layernorm, qkv + ffn_in, scales[name + ".self_attn.q_proj_input"], alpha)
        else:
            smooth_ln_fcs(layer.input_layernorm, qkv, scales[name + ".self_attn.q_proj_input"], alpha)
            smooth_ln_fcs(layer.post_attention_layernorm, ffn_in, scales[name + ".mlp.w1_input"], alpha)
        if not original_smoothquant:
            if attn.v_proj.weight.shape[0] == attn.o_proj.weight.shape[1]:       # not with grouped-query heads
                smooth_fc_fcs(attn.v_proj, attn.o_proj, scales[name + ".self_attn.o_proj_input"], alpha)
            if not original_omniquant and three:
                smooth_fc_fcs(mlp.w3, mlp.w2, scales[name + ".mlp.w2_input"], alpha)
    return model


# ---- artefacts -------------------------------------------------------------------------------------------------------------
def save_act_scales(path: str, act_scales: Dict[str, torch.Tensor]) -> None:
    """``act_scales.pth``: fp32 CPU tensors keyed "<module>_<field>" (generate_act_scale_shift.py:49-53, :170-175)."""
    torch.save({k: v.detach().float().cpu() for k, v in act_scales.items()}, path)


def load_act_scales(path: str) -> Dict[str, torch.Tensor]:
    return torch.load(path, map_location="cpu")


def save_act_dict_per_channel(path: str, act_dict: Dict[str, dict]) -> None:
    """``act_dict_per_channel.pth``: ``{module: {field: Tensor[2, C]}}`` (generate_act_range.py:155-158)."""
    torch.save({n: {f: t.detach().cpu() for f, t in fields.items()} for n, fields in act_dict.items()}, path)
