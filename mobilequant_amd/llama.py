"""Minimal llama-style decoder with the reference's LEAF-MODULE GRAPH -- nothing else of its model zoo.

MobileQuant's surgery (`create_sim_qmodel`, qmodule.py:835-865), its mixed-precision rules (ptq/mobilequant.py:175-201) and its
calibration hooks (ptq/generate_act_range.py:93-95) key on module *types* and *names*: `q_proj k_proj v_proj o_proj`, `qk_bmm
pv_bmm` (FMatMul), `w1 w2 w3`, `act_fn` (nn.SiLU), `input_layernorm post_attention_layernorm` (HFRMSNorm), final `norm`, `lm_head`
(mobilellm/model/hf_model.py:382-534, :1042-1062, :1165-1260).  This module provides exactly that graph for the shapes of
BASELINE.json (TinyLlama-1.1B: hidden 2048, 22 layers, 32 heads / 4 KV heads, head_dim 64, FFN 5632, vocab 32000 --
mobilellm/model/sim_model.py:43-44), so that the calibration benchmark (configs[4]), the layer-level benchmarks and the decode
step run on the reference's module structure with random-init weights (no checkpoints offline).  RoPE, 1/sqrt(d), mask and
softmax stay unquantised fp32 ops as in the reference (hf_model.py:486-530); residual adds and the w1*w3 product are plain ops
(not in the surgery list).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .quantization.fp_ops import FMatMul, HFRMSNorm


@dataclass
class LlamaShape:
    hidden: int = 2048
    layers: int = 22
    heads: int = 32
    kv_heads: int = 4
    head_dim: int = 64
    ffn: int = 5632
    vocab: int = 32000
    eps: float = 1e-5
    rope_theta: float = 10000.0
    max_pos: int = 2048
    hidden_act: str = "silu"            # "silu" (llama: SwiGLU) | "gelu" (gemma-style GeGLU, erf form: transformers' ACT2FN["gelu"])
    # the switches of the reference's HFConfig (mobilellm/model/hf_config.py:101-179) its other two model families set:
    norm: str = "rmsnorm"               # "layernorm": nn.LayerNorm with bias (StableLM-2; hf_model.py:1036-1040, :1197-1203, :1440)
    qkv_bias: bool = False              # attention_bias + use_qkv_bias_only: bias on q / k / v, none on o_proj (hf_model.py:414-417)
    rotary_pct: float = 1.0             # partial_rotary_factor: RoPE on the first int(pct * head_dim) dims (hf_model.py:419, :487-500)
    embed_scale: bool = False           # normalize_embed: embeddings * hidden ** 0.5 (Gemma; hf_model.py:1555-1556)

    @property
    def rot_dim(self) -> int:
        return int(self.rotary_pct * self.head_dim)

    @classmethod
    def tinyllama(cls, **kw) -> "LlamaShape":
        return cls(**kw)

    @classmethod
    def stablelm_2_1_6b(cls, **kw) -> "LlamaShape":
        """BASELINE.json configs[2]: hidden 2048, 24 layers, 32 / 32 heads of 64, FFN 5632, LayerNorm, q / k / v bias, 25 % rotary
        (the public HF config; the reference only names the model: README.md:19, scripts/convert_ckpt.py:29)."""
        base = dict(hidden=2048, layers=24, heads=32, kv_heads=32, head_dim=64, ffn=5632, vocab=100352, eps=1e-5, max_pos=4096,
                    norm="layernorm", qkv_bias=True, rotary_pct=0.25)
        base.update(kw)
        return cls(**base)

    @classmethod
    def gemma_2b(cls, **kw) -> "LlamaShape":
        """BASELINE.json configs[3] (mobilellm/model/sim_model.py:45-46): hidden 2048, 18 layers, 8 heads / 1 KV head of 256, FFN 16384,
        GeGLU, vocab 256000, scaled embeddings.  (Gemma's `1 + weight` norm is folded into the weights at checkpoint conversion:
        scripts/convert_ckpt.py:48-55, so the graph carries a plain HFRMSNorm.)"""
        base = dict(hidden=2048, layers=18, heads=8, kv_heads=1, head_dim=256, ffn=16384, vocab=256000, eps=1e-6, max_pos=8192,
                    hidden_act="gelu", embed_scale=True)
        base.update(kw)
        return cls(**base)

    @classmethod
    def toy(cls, **kw) -> "LlamaShape":
        base = dict(hidden=128, layers=2, heads=4, kv_heads=2, head_dim=32, ffn=256, vocab=97, max_pos=128)
        base.update(kw)
        return cls(**base)


def rope_tables(shape: LlamaShape, device=None):
    """cos / sin [max_pos, rot_dim] (rotate-half convention, hf_model.py:486-501; rot_dim < head_dim: partial rotary)."""
    rd = shape.rot_dim
    inv = 1.0 / (shape.rope_theta ** (torch.arange(0, rd, 2, dtype=torch.float32, device=device) / rd))
    ang = torch.outer(torch.arange(shape.max_pos, dtype=torch.float32, device=device), inv)
    ang = torch.cat((ang, ang), dim=-1)
    return ang.cos(), ang.sin()


def apply_rope(x, cos, sin):
    """x [B, H, S, D]; cos / sin [S, rot]: the first rot dims rotate, the rest passes through (hf_model.py:487-500)."""
    rd = cos.shape[-1]
    if rd < x.shape[-1]:
        return torch.cat((apply_rope(x[..., :rd], cos, sin), x[..., rd:]), dim=-1)
    h = x.shape[-1] // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    return x * cos + rot * sin


class Attention(nn.Module):
    _mq_calibration_aware = True        # forward() hands its score chain to an attached ActRangeCollector (calibration.py)
    _mq_calibration_layer_parts = ("q_proj", "k_proj", "qk_bmm", "v_proj")      # ... and RoPE with the four statistics around it (see forward)

    def __init__(self, s: LlamaShape):
        super().__init__()
        self.s = s
        self.q_proj = nn.Linear(s.hidden, s.heads * s.head_dim, bias=s.qkv_bias)
        self.k_proj = nn.Linear(s.hidden, s.kv_heads * s.head_dim, bias=s.qkv_bias)
        self.v_proj = nn.Linear(s.hidden, s.kv_heads * s.head_dim, bias=s.qkv_bias)
        self.o_proj = nn.Linear(s.heads * s.head_dim, s.hidden, bias=False)
        self.qk_bmm, self.pv_bmm = FMatMul(), FMatMul()

    def forward(self, x, cos, sin, mask, cache=None, pos: int = 0):
        """x [B, S, hidden].  cache: optional (k [B, KV, T, D], v [B, KV, T, D]) static buffers; the S new positions are
        written at pos .. pos+S-1 and attention runs over positions 0 .. pos+S-1 (sim_model.py:160-221's static cache)."""
        s = self.s
        B, S, _ = x.shape
        cl = self.__dict__.get("_mq_calib_layer")                   # (collector, names of q_proj / k_proj / qk_bmm) while ONE calibration pass runs
        rope_taken = kv_repeated = False
        if (cl is not None and cache is None and not torch.is_grad_enabled() and cl[0].can_fuse_layer(x) and s.head_dim % 4 == 0
                and cos.shape[-1] % 8 == 0 and cos.dtype == torch.float32):
            # calibration: RoPE on q and k in ONE launch that also takes the statistics of q_proj.output, k_proj.output, qk_bmm.input and
            # qk_bmm.input2 (the bits of apply_rope; the linears' and the matmul's hooks skip those fields)
            col, (nq, nk, nqk, nv) = cl
            for m in (self.q_proj, self.k_proj, self.v_proj):
                m.__dict__["_mq_calib_skip"] = (col, ("output",))
            try:
                ql, kl, vl = self.q_proj(x), self.k_proj(x), self.v_proj(x)
            finally:
                for m in (self.q_proj, self.k_proj, self.v_proj):
                    m.__dict__.pop("_mq_calib_skip", None)
            if all(t.is_contiguous() and t.dtype == torch.float32 for t in (ql, kl, vl)):
                # (+ v carried along and repeat_kv inside: k, v come back with every kv head repeated -- [B, heads, S, D])
                q, k, v = col.rope_qkv_pass(nq, nk, nv, nqk, ql, kl, vl, s.heads, s.kv_heads, s.head_dim, cos, sin)
                rope_taken = kv_repeated = True
            else:
                for n_, t_ in ((nq, ql), (nk, kl), (nv, vl)):
                    col._update(n_, "output", t_)
                q = apply_rope(ql.view(B, S, s.heads, s.head_dim).transpose(1, 2), cos, sin)
                k = apply_rope(kl.view(B, S, s.kv_heads, s.head_dim).transpose(1, 2), cos, sin)
                v = vl.view(B, S, s.kv_heads, s.head_dim).transpose(1, 2)
        else:
            q = self.q_proj(x).view(B, S, s.heads, s.head_dim).transpose(1, 2)
            k = self.k_proj(x).view(B, S, s.kv_heads, s.head_dim).transpose(1, 2)
            v = self.v_proj(x).view(B, S, s.kv_heads, s.head_dim).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        if cache is not None:
            cache[0][:, :, pos:pos + S] = k
            cache[1][:, :, pos:pos + S] = v
            k, v = cache[0][:, :, :pos + S], cache[1][:, :, :pos + S]
        rep = s.heads // s.kv_heads
        if rep > 1 and not kv_repeated:   # repeat_kv (hf_model.py:509-510)
            k = k[:, :, None].expand(B, s.kv_heads, rep, k.shape[-2], s.head_dim).reshape(B, s.heads, -1, s.head_dim)
            v = v[:, :, None].expand(B, s.kv_heads, rep, v.shape[-2], s.head_dim).reshape(B, s.heads, -1, s.head_dim)
        from .quantization import qmodule as Q
        fused = Q.attention_probs_for_training(self.qk_bmm, self.pv_bmm, q, k.transpose(2, 3), mask, math.sqrt(s.head_dim))
        if fused is not None and fused[1]:    # training: the score-sized chain between the two matmuls as one pass per direction
            pv = self.pv_bmm
            out = Q._apply(pv.output_quantizer, torch.matmul(fused[0], Q._apply(pv.input2_quantizer, v)))
            return self.o_proj(out.transpose(1, 2).reshape(B, S, s.heads * s.head_dim))
        calib = self.__dict__.get("_mq_calib")            # set by ActRangeCollector.attach() while a calibration pass runs
        if (calib is not None and fused is None and not torch.is_grad_enabled()
                and calib[0].can_fuse_attention((B, s.heads, S, k.shape[-2]), q.dtype, q.device, mask)):
            # calibration: the two [heads, S, T]-sized statistics (qk_bmm.output, pv_bmm.input) are taken inside the ONE pass that turns
            # the raw scores into probabilities, in place (calibration.ActRangeCollector.attention_probs); the module hooks skip them
            # (owner, fields): only the OWNING collector's hooks skip them -- another collector on the same model keeps its plain hooks)
            taken = ("output", "input", "input2") if (rope_taken and cl[0] is calib[0]) else ("output",)
            # (rope_taken implies cache is None: the repeated v holds exactly v_proj's output values -- pv_bmm.input2 mirrors that slot)
            pv_taken = ("input", "input2") if (rope_taken and cl[0] is calib[0] and cl[0].mirror_values(calib[2], "input2", cl[1][3], "output")) else ("input",)
            self.qk_bmm.__dict__["_mq_calib_skip"], self.pv_bmm.__dict__["_mq_calib_skip"] = (calib[0], taken), (calib[0], pv_taken)
            try:
                raw = self.qk_bmm(q, k.transpose(2, 3))
                att = calib[0].attention_probs(calib[1], calib[2], raw if raw.is_contiguous() else raw.contiguous(), mask, math.sqrt(s.head_dim))
                out = self.pv_bmm(att, v)
            finally:
                self.qk_bmm.__dict__.pop("_mq_calib_skip", None)
                self.pv_bmm.__dict__.pop("_mq_calib_skip", None)
            return self.o_proj(out.transpose(1, 2).reshape(B, S, s.heads * s.head_dim))
        if rope_taken and fused is None:
            self.qk_bmm.__dict__["_mq_calib_skip"] = (cl[0], ("input", "input2"))
            try:
                att = self.qk_bmm(q, k.transpose(2, 3)) / math.sqrt(s.head_dim)
            finally:
                self.qk_bmm.__dict__.pop("_mq_calib_skip", None)
        else:
            att = (fused[0] if fused is not None else self.qk_bmm(q, k.transpose(2, 3))) / math.sqrt(s.head_dim)
        if mask is not None:
            att = att + mask
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        out = self.pv_bmm(att, v)
        return self.o_proj(out.transpose(1, 2).reshape(B, S, s.heads * s.head_dim))


def _fused_attention_forward(self, x, cos, sin, mask, cache=None, pos: int = 0, resid=None):
    """Attention.forward with RoPE, both QMatMuls, 1/sqrt(d), the causal mask and the softmax in ONE pair of launches
    (ops.attention_quant: integer q.k^T and p.v on the MFMA units, no [S, S] tensor in memory).  Serves causal prefill (from position 0, or continuing an ImageCache at a multiple of 64)
    with static per-tensor grids (8-bit q / k / v, <= 16-bit probabilities) at head_dim 64 or 256; everything else -- training, decode steps,
    a custom mask, other shapes -- runs the module chain.  Installed by fuse_attention()."""
    from . import ops
    from .quantization import qmodule as Q
    s = self.s
    plain0 = self._mq_plain_forward

    def plain(t, *a):
        if isinstance(cache, ImageCache):
            raise RuntimeError("mobilequant_amd: an image cache (new_image_cache) only serves the fused attention -- causal chunks at "
                               "positions that are multiples of 64, static per-tensor grids, head_dim 64 / 256, no gradients")
        out = plain0(Q._materialize(t), *a)
        return out if resid is None else resid + out
    qk, pv = self.qk_bmm, self.pv_bmm
    B, S, _ = x.shape
    if isinstance(cache, ImageCache):
        # the kernel pads a chunk to a multiple of 64 rows and writes the padded K / vT rows into the cache: only the LAST chunk of a
        # sequence may be ragged, and a chunk may only go where the previous one ended -- anything else would attend pad rows as keys
        if pos != cache.filled:
            raise RuntimeError(f"mobilequant_amd: image cache holds {cache.filled} positions, the chunk was fed at pos={pos} "
                               "(chunks go in order, each where the previous one ended; new_image_cache() starts a new sequence)")
        if pos % 64:
            raise RuntimeError(f"mobilequant_amd: the previous chunk ended at position {pos}, not a multiple of 64: only the final chunk "
                               "of a sequence may be ragged (its pad rows sit in the image cache behind it)")
        if S < 2:
            raise RuntimeError("mobilequant_amd: a one-token chunk is not served by the fused attention; feed it together with the "
                               "chunk before it (65 tokens: one chunk of 65, not 64 + 1)")
    if (getattr(self, "fused_mode", "auto") == "off" or not x.is_cuda or x.dtype != torch.float32 or s.head_dim not in (64, 128, 256) or S < 2
            or (pos != 0 and not isinstance(cache, ImageCache)) or pos % 64 or not getattr(mask, "_mq_causal", False)
            or not isinstance(qk, Q.QMatMul) or not isinstance(pv, Q.QMatMul)
            or Q._needs_grad(x, *self.parameters())):
        return plain(x, cos, sin, mask, cache, pos)
    if not (Q._u8_grid(qk.input_quantizer) and Q._u8_grid(qk.input2_quantizer) and Q._u8_grid(pv.input2_quantizer)
            and Q._static_per_tensor(pv.input_quantizer, 16) and pv.input_quantizer.qmin == 0):
        return plain(x, cos, sin, mask, cache, pos)
    grids = {}
    for name, quantizer in (("qk_a", qk.input_quantizer), ("qk_b", qk.input2_quantizer), ("qk_out", qk.output_quantizer),
                            ("pv_a", pv.input_quantizer), ("pv_b", pv.input2_quantizer), ("pv_out", pv.output_quantizer)):
        g = Q.QRMSNorm._grid_or_none(quantizer)
        if g is False:
            return plain(x, cos, sin, mask, cache, pos)
        if g is not None and g[0].device != x.device:
            quantizer.scale.data, quantizer.offset.data = quantizer.scale.to(x.device), quantizer.offset.to(x.device)
            g = Q.QRMSNorm._grid_or_none(quantizer)
        grids[name] = g
    img = cache if isinstance(cache, ImageCache) else None      # chunked prefill: K / vT images of the earlier chunks
    if img is not None:
        cache = None
        if len(img.per_sequence) != B or pos + S > img.per_sequence[0]["rows"]:
            raise RuntimeError("mobilequant_amd: image cache built for another batch size or too short for this position")
        img.filled = pos + S
    akw = [dict(head_dim=s.head_dim) if img is None else dict(head_dim=s.head_dim, cache=img.per_sequence[b], pos0=pos) for b in range(B)]
    fused_qkv = _qkv_indices(self, x) if cache is None and getattr(self, "fuse_qkv", True) else None
    if fused_qkv is not None:
        idx, in_grids = fused_qkv                    # uint8 [B*S, (H + 2 KV) * 64]: one GEMM, three output grids
        idx = idx.view(B, S, -1)
        q = k = v = [None] * B
        qkv = [(idx[b], in_grids) for b in range(B)]
        batched, batched_idx = (None, None, None), (idx, in_grids)
    else:
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        qkv = [None] * B
        batched, batched_idx = (q, k, v), None
    D = s.head_dim
    if cache is not None:
        cache[0][:, :, :S] = apply_rope(k.view(B, S, s.kv_heads, D).transpose(1, 2), cos, sin)
        cache[1][:, :, :S] = v.view(B, S, s.kv_heads, D).transpose(1, 2)
    oq, o_proj = pv.output_quantizer, self.o_proj
    if Q._u8_grid(oq) and isinstance(o_proj, Q.QLinear) and not o_proj.use_temporary_parameter and o_proj.input_chan_scale is None:
        # o_proj reads pv_bmm's output grid: hand it the int8 image (fragment-blocked, + row sums) straight from the attention
        # kernel -- no fp32 [B, S, hidden] tensor, no quantize launch.  The probe tensor is never written nor read.
        M, K = B * S, s.heads * D
        probe = Q._tag_grid(torch.empty(1, dtype=torch.float32, device=x.device).expand(B, S, K), oq)
        w_o = o_proj._effective_weight(o_proj.weight)
        # _int8_ready FIRST: _weight_plan quantizes the weight, which only the integer path's configurations allow (per-group,
        # 16-bit, bypassed or absent weight quantizers fall through to o_proj(out) below, untouched)
        ready = o_proj.input_quantizer is None and M > 8 and o_proj._int8_ready(probe, w_o) and o_proj._activation_grid(probe) is oq
        w4o = ready and o_proj._weight_plan(w_o)["w4"]        # packed-only 4-bit weights: served from the image by the tiled residual kernel
        resid_tiled = (ready and resid is not None and resid.dtype == torch.float32 and resid.is_contiguous()
                       and o_proj._tiled_residual_ok(M, w_o.shape[0], K))
        if ready and (not w4o or resid_tiled):
            tiled = (not w4o and ops.gemm_tiled_supported(M, w_o.shape[0], K)) or resid_tiled
            q_i8 = torch.empty(((M + 15) // 16 * 16 if tiled else M, K), dtype=torch.int8, device=x.device)
            rs = torch.empty(M, dtype=torch.int32, device=x.device)
            if img is None and B > 1:                # the whole batch in one launch pair (sequence b owns image rows b * S ...)
                ops.attention_quant(*batched, cos, sin, s.heads, s.kv_heads, grids, image=(q_i8, rs, 0, 128, tiled), want_out=False,
                                    qkv_idx=batched_idx, head_dim=s.head_dim)
            else:
                for b in range(B):
                    ops.attention_quant(q[b], k[b], v[b], cos, sin, s.heads, s.kv_heads, grids, image=(q_i8, rs, b * S, 128, tiled), want_out=False,
                                        qkv_idx=qkv[b], **akw[b])
            return o_proj._int8_from_image(None, w_o, o_proj.bias, oq, q_i8, rs, 128, M if tiled else None, lead_shape=(B, S), resid=resid)
    if img is None and B > 1:
        out = ops.attention_quant(*batched, cos, sin, s.heads, s.kv_heads, grids, qkv_idx=batched_idx, head_dim=s.head_dim)
    else:
        out = torch.stack([ops.attention_quant(q[b], k[b], v[b], cos, sin, s.heads, s.kv_heads, grids, qkv_idx=qkv[b], **akw[b]) for b in range(B)])
    if oq is not None and not oq.bypassed():
        Q._tag_grid(out, oq)
    out = o_proj(out)
    return out if resid is None else resid + out


def _qkv_indices(self, x):
    """q_proj | k_proj | v_proj as ONE int8 GEMM with three output grids (ops.int8_linear_segmented) -> (uint8 indices
    [rows, (H + 2 KV) * 64], their (scale, offset) grids), or None when the three linears cannot share a launch (then each runs on its
    own).  Per column the index is exactly what that linear's own GEMM writes; the concatenated operands are cached on the block."""
    from . import ops
    from .quantization import qmodule as Q
    lins = (self.q_proj, self.k_proj, self.v_proj)
    if not all(isinstance(m, Q.QLinear) and not m.use_temporary_parameter and m.input_chan_scale is None and m.input_quantizer is None
               and Q._u8_grid(m.output_quantizer) for m in lins):
        return None
    ws = [m._effective_weight(m.weight) for m in lins]
    if not all(m._int8_ready(x, w) for m, w in zip(lins, ws)) or len({m.weight_quantizer.qcfg.bitwidth for m in lins}) != 1:
        return None
    grids_in = [m._activation_grid(x) for m in lins]
    if any(g.grid_token() != grids_in[0].grid_token() for g in grids_in) or x.numel() // x.shape[-1] <= 8:
        return None
    if (any(m.bias is not None for m in lins) and not all(m.bias is not None for m in lins)):
        return None
    K = ws[0].shape[1]
    M = x.numel() // x.shape[-1]
    # the fragment-blocked image a fused norm left for this block (the decoder-layer pass asks for it when the 128-column generated
    # kernel serves q | k | v): no row-major image exists then
    grid = grids_in[0]
    t_shift = 128 if grid.qmax > 127 else 0
    t_hit = None
    n_all = sum(w.shape[0] for w in ws)
    all_w4 = all(m._weight_plan(w)["w4"] for m, w in zip(lins, ws)) and n_all % 128 == 0 and ops.gemm_tiled_w4_supported(M, n_all, K)
    if (all(not m._weight_plan(w)["w4"] for m, w in zip(lins, ws)) and ops.gemm_tiled128_supported(M, n_all, K)) or all_w4:
        t_hit = Q._shared_activation.get(x, grid, ("tiled", t_shift, None))
    if t_hit is not None:
        (a_q, a_rs, a_shift), tiled_rows = t_hit, M
    else:
        grid, a_q, a_rs, a_shift, tiled_rows, decode = lins[0]._input_image(x, ws[0])
        if tiled_rows is not None or decode:
            return None
    plans = [m._epilogue_vectors(m._weight_plan(w), grid, a_shift, K) for m, w in zip(lins, ws)]
    key = tuple((p["key"], p["epi_key"]) for p in plans) + tuple(None if m.bias is None else (m.bias.data_ptr(), Q._ver(m.bias)) for m in lins)
    cat = getattr(self, "_qkv_cat", None)
    if cat is None or cat["key"] != key:
        cat = {"key": key, "w": torch.cat([p["w"] for p in plans]), "alpha": torch.cat([p["alpha"] for p in plans]),
               "w_zp": torch.cat([p["w_zp"] for p in plans]), "col_term": torch.cat([p["col_term"] for p in plans]),
               "bias": None if lins[0].bias is None else torch.cat([m.bias.detach().float() for m in lins])}
        self._qkv_cat = cat
    ends, tot = [], 0
    for w in ws:
        tot += w.shape[0]
        ends.append(tot)
    out_grids = []
    for m in lins:
        oq = m.output_quantizer
        if oq.scale.device != x.device:
            oq.scale.data, oq.offset.data = oq.scale.to(x.device), oq.offset.to(x.device)
        out_grids.append((oq.scale.detach(), oq.offset.detach()))
    if all_w4 and tiled_rows is not None:       # packed 4-bit weights on the generated kernels (QLinear.w4_prefill = "packed")
        idx = ops.w4a8_linear_tiled(a_q, M, cat["w"], a_rs, cat["alpha"], cat["w_zp"], cat["col_term"], cat["bias"], out_grids, seg_ends=ends)
        return idx, out_grids
    idx = ops.int8_linear_segmented(a_q, cat["w"], a_rs, cat["alpha"], cat["w_zp"], cat["col_term"], cat["bias"], ends, out_grids,
                                    w4=plans[0]["w4"], a_tiled_rows=tiled_rows)
    return idx, out_grids


def fuse_attention(model) -> int:
    """Graph pass (no counterpart in the reference, like quantization.fuse_gated_mlp): every Attention block whose qk_bmm / pv_bmm
    became QMatMuls gets the fused forward above.  `block.fused_mode = "off"` restores the chain.  Returns the number fused."""
    import types
    from .quantization import qmodule as Q
    n = 0
    for m in model.modules():
        if (isinstance(m, Attention) and isinstance(m.qk_bmm, Q.QMatMul) and isinstance(m.pv_bmm, Q.QMatMul)
                and not hasattr(m, "_mq_plain_forward")):
            m._mq_plain_forward = m.forward
            m.forward = types.MethodType(_fused_attention_forward, m)
            n += 1
    return n


def _fused_layer_forward(self, x, cos, sin, mask, cache=None, pos: int = 0):
    """DecoderLayer.forward with the two residual adds (ElementwiseAdd, hf_model.py:1257, :1270 -- plain fp32 adds, not in the surgery
    list) folded into the stores of the o_proj and w2 GEMMs.  Needs the fused attention and the fused gated MLP on this layer; their
    own fallbacks add the residual as a plain op, so the result is the module chain's in every case."""
    if getattr(self, "fused_mode", "auto") == "off" or not hasattr(self.self_attn, "_mq_plain_forward") or not hasattr(self.mlp, "_mq_plain_forward"):
        return self._mq_plain_forward(x, cos, sin, mask, cache, pos)
    from .quantization import qmodule as Q

    def norm(mod, t, layout):
        return mod.forward_images(t, layout) if isinstance(mod, (Q.QRMSNorm, Q.QLayerNorm)) else mod(t)
    # the norms' only consumers here are integer linears: they write just the int8 image those read (q/k/v: row-major; w1/w3: the
    # fragment-blocked layout), not the fp32 tensor
    x = self.self_attn(norm(self.input_layernorm, x, _qkv_image_layout(self.self_attn, x)), cos, sin, mask, cache, pos, resid=x)
    return self.mlp(norm(self.post_attention_layernorm, x, "tiled"), resid=x)


def _qkv_image_layout(attn, x):
    """Which int8 image the input norm should write for this attention block: the fragment-blocked one when q | k | v run as the
    segmented GEMM on the 128-column generated kernel (int8 weight images, N % 128 == 0, K % 256 == 0), else row-major."""
    from . import ops
    from .quantization import qmodule as Q
    lins = (attn.q_proj, attn.k_proj, attn.v_proj)
    if not getattr(attn, "fuse_qkv", True) or not all(isinstance(m, Q.QLinear) and m.weight_quantizer is not None for m in lins):
        return "rowmajor"
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    n = sum(m.weight.shape[0] for m in lins)
    return "tiled" if M > 8 and K % 64 == 0 and ops.gemm_tiled128_supported(M, n, K) else "rowmajor"


def fuse_decoder_layer(model) -> int:
    """fuse_attention + quantization.fuse_gated_mlp + the residual adds inside the o_proj / w2 GEMM stores, on every DecoderLayer.
    Returns the number of layers fused; `layer.fused_mode = "off"` restores the plain layer forward."""
    import types
    from .quantization import qmodule as Q
    fuse_attention(model)
    Q.fuse_gated_mlp(model)
    n = 0
    for m in model.modules():
        if (isinstance(m, DecoderLayer) and hasattr(m.self_attn, "_mq_plain_forward") and hasattr(m.mlp, "_mq_plain_forward")
                and not hasattr(m, "_mq_plain_forward")):
            m._mq_plain_forward = m.forward
            m.forward = types.MethodType(_fused_layer_forward, m)
            n += 1
    return n


class MLP(nn.Module):
    def __init__(self, s: LlamaShape):
        super().__init__()
        self.w1 = nn.Linear(s.hidden, s.ffn, bias=False)
        self.w2 = nn.Linear(s.ffn, s.hidden, bias=False)
        self.w3 = nn.Linear(s.hidden, s.ffn, bias=False)
        self.act_fn = nn.SiLU() if s.hidden_act == "silu" else nn.GELU()

    _mq_calibration_layer_parts = ("w1", "w3", "act_fn", "w2")     # calibration.ActRangeCollector.attach(): see forward

    def forward(self, x):      # hf_model.py:1057
        calib = self.__dict__.get("_mq_calib_layer")                # (collector, names of w1 / w3 / act_fn / w2) while ONE calibration pass runs
        if (calib is not None and not torch.is_grad_enabled() and calib[0].can_fuse_layer(x)
                and (type(self.act_fn) is nn.SiLU or (type(self.act_fn) is nn.GELU and self.act_fn.approximate == "none"))):
            # calibration: act(w1(x)) * w3(x) and the four statistics around it (w1.output = act.input, act.output, w3.output, w2.input)
            # in ONE pass; the linears' hooks skip those fields, the activation module is not run
            col, (n1, n3, na, n2) = calib
            w2_fields = ("input", "output") if self.__dict__.get("_mq_calib_w2_out_taken") else ("input",)      # (DecoderLayer: the next norm pass takes it)
            skips = ((self.w1, ("output",)), (self.w3, ("output",)), (self.w2, w2_fields))
            for m, f in skips:
                m.__dict__["_mq_calib_skip"] = (col, f)
            try:
                a, b = self.w1(x), self.w3(x)
                if a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float32 and a.numel() % 4 == 0:
                    return self.w2(col.gated_pass(n1, n3, na, n2, a, b, "silu" if type(self.act_fn) is nn.SiLU else "gelu"))
                for m, _ in skips:                                  # (not served: the plain chain, with the hooks' own reductions)
                    m.__dict__.pop("_mq_calib_skip", None)
                col._update(n1, "output", a)
                col._update(n3, "output", b)
                if len(w2_fields) == 2:
                    self.w2.__dict__["_mq_calib_skip"] = (col, ("output",))
                return self.w2(self.act_fn(a) * b)
            finally:
                for m, _ in skips:
                    m.__dict__.pop("_mq_calib_skip", None)
        return self.w2(self.act_fn(self.w1(x)) * self.w3(x))


def _make_norm(s: LlamaShape) -> nn.Module:
    assert s.norm in ("rmsnorm", "layernorm"), s.norm
    return HFRMSNorm(s.hidden, eps=s.eps) if s.norm == "rmsnorm" else nn.LayerNorm(s.hidden, eps=s.eps)


class DecoderLayer(nn.Module):
    def __init__(self, s: LlamaShape):
        super().__init__()
        self.self_attn = Attention(s)
        self.mlp = MLP(s)
        self.input_layernorm = _make_norm(s)
        self.post_attention_layernorm = _make_norm(s)

    _mq_calibration_layer_parts = ("input_layernorm", "post_attention_layernorm", "self_attn.o_proj", "mlp.w2")

    def forward(self, x, cos, sin, mask, cache=None, pos: int = 0, pending=None, defer: bool = False):
        """pending = (tensor, slot key, collector): a residual branch the previous layer left unadded (calibration only, see below);
        defer: this layer may do the same -- it then returns (h, (mlp output, slot key, collector)) instead of h + mlp output."""
        calib = self.__dict__.get("_mq_calib_layer")                # (collector, names of the two norms, o_proj, w2) while ONE calibration pass runs
        if (calib is not None and not torch.is_grad_enabled() and calib[0].can_fuse_layer(x)
                and calib[0].norm_is_plain(self.input_layernorm) and calib[0].norm_is_plain(self.post_attention_layernorm)):
            # calibration: each norm with its input and output statistics in one pass, the residual add in front of it inside (h = x +
            # branch is written for the next add) together with the branch's own statistic (o_proj's / w2's output hook skips it); the
            # norm modules are not run.  The layer's last add is left to the next layer's first pass (defer).
            col, (n_in, n_post, n_o, n_w2) = calib
            if pending is not None and pending[2] is col:
                x, y = col.norm_pass(n_in, self.input_layernorm, x, pending[0], pending[1])
            elif pending is not None:
                pending[2]._update(pending[1][0], pending[1][1], pending[0])
                x = x + pending[0]
                _, y = col.norm_pass(n_in, self.input_layernorm, x)
            else:
                _, y = col.norm_pass(n_in, self.input_layernorm, x)
            o = self.self_attn.o_proj
            o.__dict__["_mq_calib_skip"] = (col, ("output",))
            try:
                attn = self.self_attn(y, cos, sin, mask, cache, pos)
            finally:
                o.__dict__.pop("_mq_calib_skip", None)
            h, y = col.norm_pass(n_post, self.post_attention_layernorm, x, attn if attn.is_contiguous() else attn.contiguous(), (n_o, "output"))
            if not defer:
                return h + self.mlp(y)
            self.mlp.__dict__["_mq_calib_w2_out_taken"] = True
            try:
                m = self.mlp(y)
            finally:
                self.mlp.__dict__.pop("_mq_calib_w2_out_taken", None)
            if not m.is_contiguous() or m.dtype != torch.float32:
                col._update(n_w2, "output", m)
                return h + m
            return h, (m, (n_w2, "output"), col)
        if pending is not None:                                     # (a deferred add meets a layer on the plain path: its statistic, then the add)
            pending[2]._update(pending[1][0], pending[1][1], pending[0])
            x = x + pending[0]
        x = x + self.self_attn(self.input_layernorm(x), cos, sin, mask, cache, pos)
        return x + self.mlp(self.post_attention_layernorm(x))


class ImageCache:
    """One attention block's K / vT image caches, one per sequence of the batch (LlamaForCausalLM.new_image_cache)."""

    def __init__(self, per_sequence):
        self.per_sequence = per_sequence
        self.filled = 0          # positions written so far (true length, without the pad rows of a ragged final chunk)


class LlamaForCausalLM(nn.Module):
    """embed -> layers -> norm -> lm_head.  `layers`, final `norm` and `lm_head` carry the names the surgery rules skip
    (qmodule.py:843)."""

    _mq_calibration_layer_parts = ("norm",)        # calibration.ActRangeCollector.attach(): the final norm takes the last deferred add (forward)

    def __init__(self, shape: LlamaShape, std: float = 0.02):
        super().__init__()
        self.shape = shape
        self.embed_tokens = nn.Embedding(shape.vocab, shape.hidden)
        self.layers = nn.ModuleList(DecoderLayer(shape) for _ in range(shape.layers))
        self.norm = _make_norm(shape)
        self.lm_head = nn.Linear(shape.hidden, shape.vocab, bias=False)
        cos, sin = rope_tables(shape)
        self.register_buffer("cos", cos, persistent=False)
        self.register_buffer("sin", sin, persistent=False)
        self.reset_parameters(std)

    @torch.no_grad()
    def reset_parameters(self, std: float = 0.02, seed: Optional[int] = None):
        g = None
        if seed is not None:
            g = torch.Generator(device="cpu").manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * std) if g is not None else p.normal_(0.0, std)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * std) if g is not None else p.normal_(0.0, std)
            else:
                p.fill_(1.0)

    def calibration_alias_groups(self):
        """Hooked tensors of this graph's forward that are one and the same (calibration.ActRangeCollector reduces the first member of a
        group and mirrors the others once a first pass has confirmed them): a norm's output IS the input of the linears that read it,
        w1's output IS the activation's input; pv_bmm's output and o_proj's input hold the same values (a transposed copy: same
        minimum and maximum).  Keys are (module name, field) as named_modules() of THIS module spells them."""
        out = []
        for i in range(len(self.layers)):
            p = f"layers.{i}"
            out.append([(f"{p}.input_layernorm", "output"), (f"{p}.self_attn.q_proj", "input"), (f"{p}.self_attn.k_proj", "input"),
                        (f"{p}.self_attn.v_proj", "input")])
            out.append([(f"{p}.post_attention_layernorm", "output"), (f"{p}.mlp.w1", "input"), (f"{p}.mlp.w3", "input")])
            out.append([(f"{p}.mlp.w1", "output"), (f"{p}.mlp.act_fn", "input")])
            out.append([(f"{p}.self_attn.pv_bmm", "output"), (f"{p}.self_attn.o_proj", "input")])
        out.append([("norm", "output"), ("lm_head", "input")])
        return out

    def forward(self, ids, cache=None, pos: int = 0, last_logits_only: bool = False):
        """ids [B, S] token ids.  cache: list (one per layer) of (k, v) static buffers [B, KV, T, D], see Attention.forward.
        last_logits_only: final norm + lm_head on the LAST position only (logits [B, 1, vocab]) -- a context encoding that feeds
        generation needs no other row, and at TinyLlama's size the fp32 lm_head over 2 048 positions is 2.6 of the forward's 6.8 ms."""
        B, S = ids.shape
        x = self.embed_tokens(ids)
        if self.shape.embed_scale:
            x = x * (self.shape.hidden ** 0.5)
        cos, sin = self.cos[pos:pos + S], self.sin[pos:pos + S]
        mask = None
        if S > 1:
            mask = torch.full((S, pos + S), float("-inf"), device=x.device, dtype=x.dtype).triu(pos + 1)
            mask._mq_causal = True          # lets a fused attention skip the masked key blocks instead of reading the mask
        pend = None                 # calibration: a layer's last residual add, left to the next norm pass (DecoderLayer.forward)
        for i, layer in enumerate(self.layers):
            c = None if cache is None else cache[i]
            own = "forward" not in layer.__dict__ and type(layer).forward is DecoderLayer.forward      # (not a fused / patched layer)
            if own and (pend is not None or layer.__dict__.get("_mq_calib_layer") is not None):
                out = layer(x, cos, sin, mask, c, pos, pending=pend, defer=True)
                x, pend = out if isinstance(out, tuple) else (out, None)
            else:
                if pend is not None:                                # a deferred add in front of a layer that does not take one
                    pend[2]._update(pend[1][0], pend[1][1], pend[0])
                    x, pend = x + pend[0], None
                x = layer(x, cos, sin, mask, c, pos)
        if pend is not None:
            calib = self.__dict__.get("_mq_calib_layer")            # (collector, ("norm",)): the final norm takes the last add like a layer's
            if (calib is not None and calib[0] is pend[2] and not last_logits_only and calib[0].can_fuse_layer(x)
                    and calib[0].norm_is_plain(self.norm)):
                _, y = calib[0].norm_pass(calib[1][0], self.norm, x, pend[0], pend[1])
                return self.lm_head(y)
            pend[2]._update(pend[1][0], pend[1][1], pend[0])
            x = x + pend[0]
        if last_logits_only:
            from .quantization import qmodule as Q
            x = Q._materialize(x)[:, -1:].contiguous()
        return self.lm_head(self.norm(x))

    def new_image_cache(self, batch: int, length: int, device=None):
        """Cache for CHUNKED prefill through the fused attention (llama.fuse_attention / fuse_decoder_layer): per layer and sequence the
        int8 K / vT images the attention kernel keeps (ops.attention_image_cache).  Feed chunks with `model(ids[:, a:b], cache=c, pos=a)`,
        a % 64 == 0; every chunk attends to all earlier ones.  Chunks go in order, each at the position the previous one ended (checked: the cache
        tracks its filled length); every chunk but the last has a length that is a multiple of 64, and no chunk is a single token (feed a
        trailing token together with the chunk before it).  (No counterpart in the reference: its context encoding is one forward.)"""
        from . import ops
        s = self.shape
        device = device if device is not None else self.embed_tokens.weight.device
        return [ImageCache([ops.attention_image_cache(s.kv_heads, s.head_dim, length, device) for _ in range(batch)]) for _ in self.layers]

    def new_cache(self, batch: int, length: int, device=None, dtype=torch.float32):
        s = self.shape
        device = device if device is not None else self.embed_tokens.weight.device
        return [(torch.zeros(batch, s.kv_heads, length, s.head_dim, device=device, dtype=dtype),
                 torch.zeros(batch, s.kv_heads, length, s.head_dim, device=device, dtype=dtype)) for _ in self.layers]
