"""Single-token decode with a static KV cache: MI355X-native counterpart of the per-token body of ``SimModel.generate``
(mobilellm/model/sim_model.py:160-221) for a simulated-quant llama-style model.

``DecodeEngine`` takes a ``mobilequant_amd.llama.LlamaForCausalLM`` that went through the reference's surgery
(``create_sim_qmodel`` -> ``update_qcfg`` / mixed-precision rules -> ``set_scale_and_offset``) and lowers every decoder layer to the
five fused launches of ``csrc/mq_decode.hip``:

    input_layernorm + q|k|v stream  ->  RoPE / cache append / qk_bmm / softmax / pv_bmm  ->  o_proj stream + residual
    ->  post_attention_layernorm + interleaved w1|w3 stream + QSiLU * (.) + w2's input quantizer  ->  w2 stream + residual

plus final norm + lm_head (floating point, as the surgery leaves them: qmodule.py:843).  Integer weights, epilogue vectors and
quantizer grids are taken from the Q-modules themselves (the same caches the prefill path uses), so a decode step computes what
the module graph computes for that position: each logit within the tolerance the prefill kernels state.  The token
id and the position live in device memory: ``capture()`` records ONE hipGraph that serves every step of a generation.
W8A8 (8-bit weights); 8-bit unsigned activation grids; 16-bit grids where the recipe puts them (norm inputs, o_proj / w2 outputs,
qk_bmm output, pv_bmm input).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib, ops
from ._lib import MQ_U8, MqDecodeAttentionArgs, MqDecodeAttentionOprojArgs, MqDecodeGemvArgs, MqGrid
from .quantization import qmodule as Q


def _grid(q: Optional[Q.Quantizer], keep: list) -> MqGrid:
    """Device view of a static per-tensor quantizer (absent / bypassed -> the null grid)."""
    if q is None or q.bypassed():
        return MqGrid(None, None, 0.0, 0.0)
    if not Q._static_per_tensor(q, 16):
        raise RuntimeError("DecodeEngine needs static per-tensor activation grids (set_scale_and_offset first)")
    s, o = q.scale.detach().float().contiguous(), q.offset.detach().float().contiguous()
    keep += [s, o]
    if isinstance(keep, _Keep):
        keep.sources.append((q, q.grid_token()))
    return MqGrid(s.data_ptr(), o.data_ptr(), float(q.qmin), float(q.qmax))


class _Keep(list):
    """Tensors the launch records point into, plus what they were derived from: (quantizer, grid_token) for every grid packed into a
    constants line and (weight, version) for every weight image, so that the engine can tell when the model moved under it."""

    def __init__(self):
        super().__init__()
        self.sources, self.weights = [], []

    def stale(self) -> bool:
        return any(q.grid_token() != t for q, t in self.sources) or any(Q._ver(w) != v for w, v in self.weights)


class _Linear:
    """Integer image of one or several QLinears that read the same activation grid: weights [N, K] int8 (index - 128) and the
    per-row epilogue vectors for that grid, concatenated (q|k|v) or row-interleaved (w1|w3)."""

    @staticmethod
    def _check(lin: Q.QLinear):
        """The conditions QLinear._int8_ready / the gated-MLP pass put on a linear before its integer path may stand in for the
        module: what they reject, the engine must not silently compute differently (qmodule.py:341-358 is the contract)."""
        wq, name = lin.weight_quantizer, type(lin).__name__
        if lin.int8_mode == "off":
            raise RuntimeError(f"DecodeEngine: {name}.int8_mode == 'off' asks for the simulated path")
        if wq is None or wq.bypassed() or wq.qcfg.bitwidth > 8 or wq.qcfg.is_dynamic or wq.lwc:
            raise RuntimeError("DecodeEngine: weight quantizers must be static, <= 8 bit and not in LWC mode (run the PTQ to its end first)")
        if wq.qcfg.is_per_channel and wq.qcfg.group_size != -1:
            raise RuntimeError("DecodeEngine: grouped per-channel weight quantizers are not served by the integer kernels")
        if lin.use_temporary_parameter or getattr(lin, "temp_weight", None) is not None:
            raise RuntimeError("DecodeEngine: fold the LET parameters first (smooth_lm_inplace): temp_weight / use_temporary_parameter is set")
        if lin.input_chan_scale is not None:
            raise RuntimeError("DecodeEngine: fold the run-time SmoothQuant channel scale into the weights first (smoothquant.smooth_lm)")
        for role in ("input_quantizer", "output_quantizer"):
            q = getattr(lin, role)
            if q is not None and not q.bypassed() and not Q._static_per_tensor(q, 16):
                raise RuntimeError(f"DecodeEngine: {role} must be a static per-tensor grid of at most 16 bits")

    def __init__(self, linears: List[Q.QLinear], a_grid: Q.Quantizer, interleave: bool = False):
        ws, alphas, zps, cts, biases = [], [], [], [], []
        for lin in linears:
            self._check(lin)
        bits = {lin.weight_quantizer.qcfg.bitwidth for lin in linears}
        if len(bits) != 1 or not bits <= {4, 8}:
            raise RuntimeError("DecodeEngine: the linears of one phase need the same 8- or 4-bit weight quantizer width")
        self.w4 = bits == {4}
        for lin in linears:
            K = lin.weight.shape[1]
            if self.w4 and K % 64:
                raise RuntimeError("DecodeEngine: packed 4-bit weights need K % 64 == 0")
            plan = lin._epilogue_vectors(lin._weight_plan(lin.weight), a_grid, 128, K)
            ws.append(Q.QLinear._decode_weights(plan)[0]); alphas.append(plan["alpha"].clone()); zps.append(plan["w_zp"].clone()); cts.append(plan["col_term"].clone())
            biases.append(lin.bias.detach().float() if lin.bias is not None else None)
            plan["epi_key"] = None                      # the prefill path re-derives its vectors for its own grid object
        cat = (lambda ts: torch.stack(ts, dim=1).reshape(-1, *ts[0].shape[1:])) if interleave else (lambda ts: torch.cat(ts, dim=0))
        self.w = cat(ws).contiguous()
        self.alpha, self.w_zp, self.col_term = cat(alphas).contiguous(), cat(zps).contiguous(), cat(cts).contiguous()
        self.bias = None
        if any(b is not None for b in biases):
            self.bias = cat([b if b is not None else torch.zeros(l.weight.shape[0], device=l.weight.device)
                             for b, l in zip(biases, linears)]).contiguous()
        self.N, self.K = self.w.shape[0], linears[0].weight.shape[1]
        self.rows = [l.weight.shape[0] for l in linears]
        self.sources = [(l.weight_quantizer, l.weight_quantizer.grid_token()) for l in linears]
        self.weights = [(l.weight, Q._ver(l.weight)) for l in linears]
        self._lins = list(linears)

    def byte_rows(self) -> torch.Tensor:
        """[N, K] int8, ONE byte per weight whatever the stream format: index - 128 of an 8-bit weight, the unsigned nibble of a 4-bit one
        (the numbers the packed kernels unpack: the same epilogue vectors apply)."""
        outs = []
        for lin in self._lins:
            plan = lin._weight_plan(lin.weight)
            if plan["bits4"] and plan["w4"]:                     # packed-only module (QLinear.w4_prefill = 'packed'): rebuild the nibbles
                wq = lin.weight_quantizer
                q, _ = ops.quantize(lin.weight.detach().float(), wq.scale.detach(), wq.offset.detach(), wq.qmin, wq.qmax, q_dtype=MQ_U8,
                                    shift=wq.qmin, rows=lin.weight.shape[0], want_row_sum=True)
                outs.append(q.view(torch.int8))
            else:
                outs.append(plan["w"].view(torch.int8).reshape(lin.weight.shape))
        return torch.cat(outs, dim=0).contiguous()


class DecodeEngine:
    LONG_FROM, LONG_SPLITS = 768, 4      # five launches: the split attention launch from LONG_FROM cached positions on
    LONG4_FROM = 1024                    # four launches: the 1024-thread attention + o_proj launch from here on (mq_decode_attention_oproj_args.threads):
                                         # 256 / 1024 threads at 256 | 512 | 1024 | 2048 positions: 1768 | 1701 | 1555 | 1331 against 1669 | 1646 | 1595 | 1482 tok/s

    def __init__(self, model, cache_len: int = 2048, attn_splits: Optional[int] = None, prefetch: float = 0.5, prefetch_delay_us: Optional[float] = None,
                 launches: int = 4, long_from: Optional[int] = None):
        """launches: 4 (round 6, default) = per layer {norm + q|k|v, RoPE / cache append / attention + o_proj's contraction, o_proj's
        epilogue + norm + w1|w3 + gate, w2}; 5 = round 2-5's chain with o_proj as a launch of its own.  A geometry the 4-launch kernels
        do not serve falls back to 5 (self.launches says which).  The 4-launch chain keeps the VALUE cache transposed in 16-position chunks
        ([kv_heads, cache_len / 16, head_dim, 16]: its p.v sweep is v_dot4 work on coalesced KiB requests); use cached_values() / load_cached_values() to read / write it in
        the logical [kv_heads, positions, head_dim] layout."""
        from .llama import LlamaForCausalLM
        assert isinstance(model, LlamaForCausalLM)
        assert launches in (4, 5)
        self.model, self.shape = model, model.shape
        s = self.shape
        dev = next(model.parameters()).device
        self.dev, self.cache_len = dev, int(cache_len)
        self._prefetch = (prefetch, prefetch_delay_us)           # (delay None: by chain, once self.launches is known)
        self.cos, self.sin = model.cos.contiguous(), model.sin.contiguous()
        self.oproj_geom = self._oproj_geometry(s, self.cos.shape[1]) if launches == 4 and self.cache_len % 16 == 0 else None
        self.launches = 4 if self.oproj_geom is not None else 5
        self.v_transposed = self.launches == 4
        self.long_from = self.LONG4_FROM if long_from is None else int(long_from)      # (four launches) first position of the long-cache graph
        if prefetch_delay_us is None:
            # when the prefetch rows of the attention launch start streaming w1|w3 into the L2s: behind the attention's own dependent
            # requests.  Five launches: 1.5 us (round 3).  Four launches: 2.5 us -- 0.8 / 1.5 / 2.5 -> 1 748 / 1 743 / 1 756-1 766 tok/s
            # (the attention + o_proj launch is longer and requests o_proj's weights behind its scores: profiles/r06/decode_prefetch_ab.log)
            self._prefetch = (prefetch, 2.5 if self.launches == 4 else 1.5)
        if self.launches == 4:
            self.o_acc = torch.zeros(s.hidden, dtype=torch.int32, device=dev)              # o_proj's integer sums (split-K over the heads)
            self.x_mid = torch.zeros(s.hidden, device=dev)                                 # residual stream behind the attention block
            self.rope_row = torch.zeros(2 * self.cos.shape[1], device=dev)                 # {cos[pos], sin[pos]}, staged once per token
        self.x = torch.zeros(s.hidden, device=dev)
        self.qkv = torch.zeros((s.heads + 2 * s.kv_heads) * s.head_dim, device=dev)
        self.attn_q = torch.zeros(s.heads * s.head_dim, dtype=torch.int8, device=dev)     # pv_bmm's output as o_proj's int8 image
        # workgroups per head in the attention launch (64-position blocks interleaved over them).  None = by position: one workgroup
        # per head while the cache is short (the ticket-ordered combine of a split launch costs ~3 us), LONG_SPLITS of them from
        # LONG_FROM cached positions on, where the sweeps over the cache outweigh it (measured crossover)
        self.auto_splits = attn_splits is None
        self.attn_splits = int(attn_splits) if attn_splits else 1
        assert 1 <= self.attn_splits <= 16
        self.attn_part = torch.zeros(max(self.attn_splits, self.LONG_SPLITS), s.heads * s.head_dim, dtype=torch.int64, device=dev)
        self.attn_ticket = torch.zeros(s.heads, dtype=torch.int32, device=dev)
        self.gate_q = torch.zeros(s.ffn, dtype=torch.int8, device=dev)
        self.logits = torch.zeros(s.vocab, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tok = torch.zeros(1, dtype=torch.int64, device=dev)
        # keys / values as int8 indices (index - 128) on qk_bmm.input2 / pv_bmm.input2's grids
        self.k_cache = [torch.zeros(s.kv_heads, self.cache_len, s.head_dim, dtype=torch.int8, device=dev) for _ in model.layers]
        # four launches: [kv_heads, cache_len / 16, head_dim, 16] -- 16-position chunks of all dimensions, each dimension's 16 positions
        # contiguous (the p.v sweep reads a chunk as one coalesced KiB and contracts it with v_dot4); five launches: [kv_heads, cache_len, head_dim]
        vshape = (s.kv_heads, self.cache_len // 16, s.head_dim, 16) if self.v_transposed else (s.kv_heads, self.cache_len, s.head_dim)
        self.v_cache = [torch.zeros(vshape, dtype=torch.int8, device=dev) for _ in model.layers]
        self._host_pos = 0                                   # mirror of self.pos for the cache-overflow guard (no device read-back)
        assert self.cos.shape[0] >= self.cache_len, "rope tables shorter than the cache"
        self.graph = None
        self.graph_long = None
        self._lower()

    @staticmethod
    def _oproj_geometry(s, rot_dim):
        """(slices, threads per row) of mq_decode_attention_oproj for this shape, or None when the 4-launch kernels do not serve it:
        heads x slices workgroups ~ one per CU; a workgroup's 256 threads hold hidden / slices rows x head_dim bytes of o_proj."""
        D, H, N = s.head_dim, s.heads, s.hidden
        if D not in (32, 64, 128, 256) or rot_dim % 2 or rot_dim > min(D, 256) or N > 4096 or N % 4:
            return None
        chunks = D // 16
        for slices in range(max(256 // H, 1), 0, -1):
            if N % slices:
                continue
            R = N // slices
            for tpr in (1, 2, 4):                              # fewest threads per row: the atomics leave in full waves
                if chunks % tpr == 0 and chunks // tpr <= 8 and R * tpr <= 256:
                    return slices, tpr
        return None

    def _lower(self):
        """Build the launch records from the model as it is now: weight images, epilogue vectors and one constants line per launch
        (_pack) are SNAPSHOTS of the quantizers -- the kernels do not read through the module's scale / offset tensors."""
        model, dev, s = self.model, self.dev, self.shape
        prefetch, prefetch_delay_us = self._prefetch
        self._keep = _Keep()
        self.phases = []          # (kind, ctypes struct) in launch order
        # embedding table, final norm and lm_head (fp32, unquantised: qmodule.py:843) are snapshots like every weight image: re-derived
        # here and tracked for grids_stale()
        self.embed = model.embed_tokens.weight.detach()
        if s.embed_scale:                                    # normalize_embed (Gemma; hf_model.py:1555-1556): x = embed * hidden ** 0.5
            self.embed = self.embed * (s.hidden ** 0.5)      # the same fp32 product the module graph forms per token
        self.norm_ln = isinstance(model.norm, torch.nn.LayerNorm)
        self.norm_w = model.norm.weight.detach().float().contiguous()
        self.norm_b = model.norm.bias.detach().float().contiguous() if getattr(model.norm, "bias", None) is not None else None
        self.lm_w = model.lm_head.weight.detach().float().contiguous()
        self.lm_b = model.lm_head.bias.detach().float().contiguous() if model.lm_head.bias is not None else None
        for w in (model.embed_tokens.weight, model.norm.weight, getattr(model.norm, "bias", None), model.lm_head.weight, model.lm_head.bias):
            if w is not None:
                self._keep.weights.append((w, Q._ver(w)))
        for q in model.modules():                 # grids set from act_dict.json sit on the host until a forward moves them
            if isinstance(q, Q.Quantizer) and q._has_grid() and q.scale.device != dev:
                q.scale.data, q.offset.data = q.scale.to(dev), q.offset.to(dev)
        self.oproj_images = []
        with torch.no_grad():
            for li, layer in enumerate(model.layers):
                (self._lower_layer4 if self.launches == 4 else self._lower_layer)(li, layer)
        # The attention launch of layer L pulls (a share of) layer L's w1|w3 stream into the L2s with extra workgroups: it keeps 32 of
        # 256 CUs busy and leaves the memory fabric idle, while w1|w3 is the step's biggest stream.  Measured (TinyLlama shape, context
        # 256): share 0 / 0.5 / 0.7 / 1.0 -> 0.678 / 0.656 / 0.665 / 0.690 ms per token: the attention's own dependent loads queue
        # behind the prefetch stream, so half of it, started 1.5 us into the launch, is the optimum.
        if prefetch:
            pairs = [(self.phases[i][1], self.phases[i + (1 if self.launches == 4 else 2)][1]) for i in range(1, len(self.phases), self.launches)]
            for at, gate in pairs:
                n, per, tot = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
                _lib.call("mq_decode_gemv_geometry", ctypes.byref(gate), ctypes.byref(n), ctypes.byref(per), ctypes.byref(tot))
                at.prefetch, at.prefetch_stride, at.prefetch_total, at.prefetch_wgs = gate.w, per.value, tot.value, n.value
                at.prefetch_bytes_per_wg = min(per.value, int(per.value * float(prefetch)) // 1024 * 1024)
                at.prefetch_delay = int(prefetch_delay_us * 100)
        self.weight_bytes = sum(p[1]._mq_bytes for p in self.phases if hasattr(p[1], "_mq_bytes"))
        self.head_bytes = self.lm_w.numel() * 4

    def grids_stale(self) -> bool:
        """True when a quantizer grid or a weight the engine snapshotted has been changed since (in place or replaced)."""
        return self._keep.stale()

    def refresh_grids(self):
        """Re-derive every weight image, epilogue vector and constants line from the model's current quantizers and re-record the
        graph(s) if there were any.  The cached keys / values stay as they are: they are indices on the OLD qk_bmm / pv_bmm input
        grids, so after a recalibration start the sequence again (reset() / prefill())."""
        had_graph = self.graph is not None
        self.graph = self.graph_long = None
        self._lower()
        if had_graph:
            self.capture()
        return self

    def _sync_grids(self):
        # checked where a sequence starts (capture / reset / prefill), not per step(): a token is 0.65 ms, the walk over ~500 grids
        # is about as long.  A grid changed in the middle of a sequence is the caller's to announce with refresh_grids().
        if self._keep.stale():
            self.refresh_grids()

    # -- lowering ----------------------------------------------------------------------------------------------------------
    def _norm_args(self, norm, a: MqDecodeGemvArgs):
        """QRMSNorm (qmodule.py:469-530) or QLayerNorm (qmodule.py:579-640; StableLM-2) fused in front of the weight stream."""
        ln = isinstance(norm, Q.QLayerNorm)
        if ln:
            if norm.use_temporary_parameter or norm.weight is None:
                raise RuntimeError("DecodeEngine: QLayerNorm needs its affine weight and no temporary (LET) parameters")
        elif not isinstance(norm, Q.QRMSNorm) or norm.l2norm_as_rmsnorm or norm.bias is not None:
            raise RuntimeError("DecodeEngine: QRMSNorm (plain RMS form, no bias) or QLayerNorm layers only")
        wfq = Q._apply(norm.weight_quantizer, norm.weight.detach()).float().contiguous()
        self._keep.append(wfq)
        self._keep.weights.append((norm.weight, Q._ver(norm.weight)))
        if norm.weight_quantizer is not None and norm.weight_quantizer._has_grid():
            self._keep.sources.append((norm.weight_quantizer, norm.weight_quantizer.grid_token()))
        a.norm_w, a.norm_in, a.eps = wfq.data_ptr(), _grid(norm.input_quantizer, self._keep), float(norm.eps)
        a.layernorm = int(ln)
        if ln and norm.bias is not None:
            nb = norm.bias.detach().float().contiguous()
            self._keep.append(nb)
            a.norm_bias = nb.data_ptr()
        a.a_grid = _grid(norm.output_quantizer, self._keep)
        if norm.output_quantizer is None or norm.output_quantizer.qmax != 255:
            raise RuntimeError("DecodeEngine: the norm feeding a linear needs an 8-bit unsigned output grid")
        return norm.output_quantizer

    def _gemv(self, lin: _Linear, **fields) -> MqDecodeGemvArgs:
        a = MqDecodeGemvArgs()
        a.K, a.N = lin.K, lin.N
        a.w, a.alpha, a.w_zp, a.col_term = lin.w.data_ptr(), lin.alpha.data_ptr(), lin.w_zp.data_ptr(), lin.col_term.data_ptr()
        a.bias = lin.bias.data_ptr() if lin.bias is not None else None
        a.seg_end[0] = a.seg_end[1] = lin.N
        for k, v in fields.items():
            setattr(a, k, v)
        a.w4 = int(lin.w4)
        a._mq_bytes = lin.N * lin.K // (2 if lin.w4 else 1)
        self._keep.append(lin)
        self._keep.sources += lin.sources
        self._keep.weights += lin.weights
        return a

    def _pack(self, grids) -> int:
        """mq_decode_pack_grids: the launch's static grids -> one constants line on the device (no host read-back)."""
        arr = (MqGrid * len(grids))(*grids)
        out = torch.zeros(64, device=self.dev)                 # one 256-byte line: every wave reads all 64 floats
        _lib.call("mq_decode_pack_grids", arr, len(grids), out.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream)
        self._keep.append(out)
        return out.data_ptr()

    def _finish_gemv(self, a: MqDecodeGemvArgs) -> MqDecodeGemvArgs:
        a.consts = self._pack([a.norm_in, a.a_grid, a.out_grid[0], a.out_grid[1], a.out_grid[2], a.gate_mid, a.gate_actout, a.gate_out, a.o_out])
        return a

    def _attention_grids(self, attn, at, keep):
        qk, pv = attn.qk_bmm, attn.pv_bmm
        at.qk_a, at.qk_b, at.qk_out = (_grid(q, keep) for q in (qk.input_quantizer, qk.input2_quantizer, qk.output_quantizer))
        at.pv_a, at.pv_b, at.pv_out = (_grid(q, keep) for q in (pv.input_quantizer, pv.input2_quantizer, pv.output_quantizer))
        # o_proj's input sits on pv_bmm's output grid (the live producer), else on its declared / own grid: the attention launch
        # writes pv_bmm's output straight as o_proj's int8 image on that grid
        g_o = attn.o_proj.input_quantizer if attn.o_proj.input_quantizer is not None else (
            pv.output_quantizer if Q._static_per_tensor(pv.output_quantizer, 8) else attn.o_proj._input_grid)
        if g_o is None or g_o.qmax != 255:
            raise RuntimeError("DecodeEngine: o_proj needs an 8-bit unsigned input grid (pv_bmm output)")
        at.o_in = _grid(g_o, keep)
        at.consts = self._pack([at.qk_a, at.qk_b, at.qk_out, at.pv_a, at.pv_b, at.pv_out, at.o_in])
        return g_o

    def _lower_layer4(self, li, layer):
        """Round 6: four launches per layer (csrc/mq_decode.hip: decode_attention_oproj_kernel, OPRE)."""
        s, keep = self.shape, self._keep
        attn, mlp = layer.self_attn, layer.mlp
        for m in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.w1, mlp.w2, mlp.w3):
            if not isinstance(m, Q.QLinear):
                raise RuntimeError("DecodeEngine: run create_sim_qmodel first")
        slices, tpr = self.oproj_geom
        # (1) input_layernorm + q|k|v stream (the five-launch chain's launch); also clears o_proj's sums
        a = MqDecodeGemvArgs()
        g_in = self._norm_args(layer.input_layernorm, a)
        qkv = _Linear([attn.q_proj, attn.k_proj, attn.v_proj], g_in)
        p1 = self._gemv(qkv, x=self.x.data_ptr(), norm_w=a.norm_w, norm_bias=a.norm_bias, layernorm=a.layernorm, norm_in=a.norm_in, eps=a.eps,
                        a_grid=a.a_grid, y=self.qkv.data_ptr(), zero_acc=self.o_acc.data_ptr(), zero_n=s.hidden)
        p1.seg_end[0], p1.seg_end[1] = qkv.rows[0], qkv.rows[0] + qkv.rows[1]
        for k, lin in enumerate((attn.q_proj, attn.k_proj, attn.v_proj)):
            p1.out_grid[k] = _grid(lin.output_quantizer, keep)
        self.phases.append(("gemv", self._finish_gemv(p1)))
        # (2) RoPE / cache append / attention + o_proj's contraction (split-K over the heads, exact integer atomics)
        at = MqDecodeAttentionOprojArgs()
        at.qkv, at.k_cache, at.v_cache = self.qkv.data_ptr(), self.k_cache[li].data_ptr(), self.v_cache[li].data_ptr()
        at.rope_row, at.pos = self.rope_row.data_ptr(), self.pos.data_ptr()
        at.heads, at.kv_heads, at.head_dim, at.cache_len, at.rot_dim = s.heads, s.kv_heads, s.head_dim, self.cache_len, self.cos.shape[1]
        g_o = self._attention_grids(attn, at, keep)
        op = _Linear([attn.o_proj], g_o)
        if op.K != s.heads * s.head_dim or op.N != s.hidden:
            raise RuntimeError("DecodeEngine: o_proj must map heads * head_dim -> hidden")
        o_w = op.byte_rows().view(op.N, s.heads, s.head_dim).permute(1, 0, 2).contiguous()          # [heads][N][D]: a head's K-slice of every row
        keep += [o_w, op]
        self.oproj_images.append((o_w, op))                   # (per layer; tests read them)
        keep.sources += op.sources
        keep.weights += op.weights
        at.o_w, at.o_wzp, at.o_acc, at.N, at.slices, at.tpr = o_w.data_ptr(), op.w_zp.data_ptr(), self.o_acc.data_ptr(), op.N, slices, tpr
        lg = lambda n: n.bit_length() - 1 if n > 0 and n & (n - 1) == 0 else None      # noqa: E731
        lgs, lgg, lgk = lg(slices), lg(s.heads // s.kv_heads), lg(s.kv_heads)
        ok = None not in (lgs, lgg, lgk) and (lgk > 3 or lgs + lgg >= 3 - lgk)
        at.lg_slices, at.lg_group, at.lg_kv = (lgs, lgg, lgk) if ok else (-1, 0, 0)
        at._mq_bytes = o_w.numel()
        self.phases.append(("attn_oproj", at))
        # (3) o_proj's epilogue + residual -> post_attention_layernorm + interleaved w1|w3 + gated activation + w2's input quantizer
        a2 = MqDecodeGemvArgs()
        g_ffn = self._norm_args(layer.post_attention_layernorm, a2)
        w13 = _Linear([mlp.w1, mlp.w3], g_ffn, interleave=True)
        act = mlp.act_fn
        if not isinstance(act, (Q.QSiLU, Q.QGELU)) or (act.input_quantizer is not None and not act.input_quantizer.bypassed()):
            raise RuntimeError("DecodeEngine: act_fn must be QSiLU / QGELU without an input quantizer (the reference's surgery)")
        iq2 = mlp.w2.input_quantizer
        if iq2 is None or iq2.qmax != 255:
            raise RuntimeError("DecodeEngine: w2 needs its own 8-bit unsigned input quantizer")
        p4 = self._gemv(w13, x=self.x.data_ptr(), norm_w=a2.norm_w, norm_bias=a2.norm_bias, layernorm=a2.layernorm, norm_in=a2.norm_in,
                        eps=a2.eps, a_grid=a2.a_grid,
                        gate_q=self.gate_q.data_ptr(), gate_act=0 if isinstance(act, Q.QSiLU) else 1,
                        gate_mid=_grid(act.input2_quantizer if isinstance(act, Q.QSiLU) else None, keep),
                        gate_actout=_grid(act.output_quantizer, keep), gate_out=_grid(iq2, keep),
                        o_acc=self.o_acc.data_ptr(), o_alpha=op.alpha.data_ptr(), o_ct=op.col_term.data_ptr(),
                        o_bias=op.bias.data_ptr() if op.bias is not None else None, o_out=_grid(attn.o_proj.output_quantizer, keep),
                        x_mid=self.x_mid.data_ptr())
        p4.out_grid[0], p4.out_grid[1] = _grid(mlp.w1.output_quantizer, keep), _grid(mlp.w3.output_quantizer, keep)
        self.phases.append(("gemv", self._finish_gemv(p4)))
        # (4) w2 from the int8 image + residual (the stream behind the attention block)
        w2 = _Linear([mlp.w2], iq2)
        p5 = self._gemv(w2, xq=self.gate_q.data_ptr(), a_grid=_grid(iq2, keep), resid=self.x_mid.data_ptr(), y=self.x.data_ptr())
        p5.out_grid[0] = _grid(mlp.w2.output_quantizer, keep)
        self.phases.append(("gemv", self._finish_gemv(p5)))

    def _lower_layer(self, li, layer):
        phases = self.phases
        s, keep = self.shape, self._keep
        attn, mlp = layer.self_attn, layer.mlp
        for m in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.w1, mlp.w2, mlp.w3):
            if not isinstance(m, Q.QLinear):
                raise RuntimeError("DecodeEngine: run create_sim_qmodel first")
        # (1) input_layernorm + q|k|v
        a = MqDecodeGemvArgs()
        g_in = self._norm_args(layer.input_layernorm, a)
        qkv = _Linear([attn.q_proj, attn.k_proj, attn.v_proj], g_in)
        p1 = self._gemv(qkv, x=self.x.data_ptr(), norm_w=a.norm_w, norm_bias=a.norm_bias, layernorm=a.layernorm, norm_in=a.norm_in, eps=a.eps,
                        a_grid=a.a_grid, y=self.qkv.data_ptr())
        p1.seg_end[0], p1.seg_end[1] = qkv.rows[0], qkv.rows[0] + qkv.rows[1]
        for k, lin in enumerate((attn.q_proj, attn.k_proj, attn.v_proj)):
            p1.out_grid[k] = _grid(lin.output_quantizer, keep)
        phases.append(("gemv", self._finish_gemv(p1)))
        # (2) attention core
        at = MqDecodeAttentionArgs()
        at.qkv, at.k_cache, at.v_cache = self.qkv.data_ptr(), self.k_cache[li].data_ptr(), self.v_cache[li].data_ptr()
        at.cos, at.sin, at.pos = self.cos.data_ptr(), self.sin.data_ptr(), self.pos.data_ptr()
        at.heads, at.kv_heads, at.head_dim, at.cache_len = s.heads, s.kv_heads, s.head_dim, self.cache_len
        at.rot_dim, at.nsplit = self.cos.shape[1], self.attn_splits
        g_o = self._attention_grids(attn, at, keep)
        at.out_q, at.part, at.ticket = self.attn_q.data_ptr(), self.attn_part.data_ptr(), self.attn_ticket.data_ptr()
        phases.append(("attn", at))
        # (3) o_proj + residual from the int8 image
        op = _Linear([attn.o_proj], g_o)
        p3 = self._gemv(op, xq=self.attn_q.data_ptr(), a_grid=_grid(g_o, keep), resid=self.x.data_ptr(), y=self.x.data_ptr())
        p3.out_grid[0] = _grid(attn.o_proj.output_quantizer, keep)
        phases.append(("gemv", self._finish_gemv(p3)))
        # (4) post_attention_layernorm + interleaved w1|w3 + gated activation + w2's input quantizer
        a2 = MqDecodeGemvArgs()
        g_ffn = self._norm_args(layer.post_attention_layernorm, a2)
        w13 = _Linear([mlp.w1, mlp.w3], g_ffn, interleave=True)
        act = mlp.act_fn
        if not isinstance(act, (Q.QSiLU, Q.QGELU)) or (act.input_quantizer is not None and not act.input_quantizer.bypassed()):
            raise RuntimeError("DecodeEngine: act_fn must be QSiLU / QGELU without an input quantizer (the reference's surgery)")
        iq2 = mlp.w2.input_quantizer
        if iq2 is None or iq2.qmax != 255:
            raise RuntimeError("DecodeEngine: w2 needs its own 8-bit unsigned input quantizer")
        p4 = self._gemv(w13, x=self.x.data_ptr(), norm_w=a2.norm_w, norm_bias=a2.norm_bias, layernorm=a2.layernorm, norm_in=a2.norm_in,
                        eps=a2.eps, a_grid=a2.a_grid,
                        gate_q=self.gate_q.data_ptr(), gate_act=0 if isinstance(act, Q.QSiLU) else 1,
                        gate_mid=_grid(act.input2_quantizer if isinstance(act, Q.QSiLU) else None, keep),
                        gate_actout=_grid(act.output_quantizer, keep), gate_out=_grid(iq2, keep))
        p4.out_grid[0], p4.out_grid[1] = _grid(mlp.w1.output_quantizer, keep), _grid(mlp.w3.output_quantizer, keep)
        phases.append(("gemv", self._finish_gemv(p4)))
        # (5) w2 from the int8 image + residual
        w2 = _Linear([mlp.w2], iq2)
        p5 = self._gemv(w2, xq=self.gate_q.data_ptr(), a_grid=_grid(iq2, keep), resid=self.x.data_ptr(), y=self.x.data_ptr())
        p5.out_grid[0] = _grid(mlp.w2.output_quantizer, keep)
        phases.append(("gemv", self._finish_gemv(p5)))

    # -- running -------------------------------------------------------------------------------------------------------------
    _ENTRY = {"gemv": "mq_decode_gemv", "attn": "mq_decode_attention", "attn_oproj": "mq_decode_attention_oproj"}

    def _launch(self, phases=None):
        """embedding gather + 4 (or 5) launches per layer + norm / lm_head, on the current stream; reads self.tok / self.pos."""
        st = torch.cuda.current_stream(self.dev).cuda_stream
        phases = self.phases if phases is None else phases
        if phases and phases[1][0] == "attn_oproj":             # token start: embedding row + this position's cos / sin row
            _lib.call("mq_decode_embed", self.embed.data_ptr(), self.tok.data_ptr(), self.shape.hidden, self.embed.shape[0], self.cos.data_ptr(),
                      self.sin.data_ptr(), self.pos.data_ptr(), self.cos.shape[1], self.cos.shape[0], self.x.data_ptr(), self.rope_row.data_ptr(), st)
        else:
            torch.index_select(self.embed, 0, self.tok, out=self.x.view(1, -1))
        for kind, a in phases:
            _lib.call(self._ENTRY[kind], ctypes.byref(a), st)
        _lib.call("mq_decode_head", self.x.data_ptr(), self.norm_w.data_ptr(), self.norm_b.data_ptr() if self.norm_b is not None else None,
                  int(self.norm_ln), float(self.model.norm.eps), self.lm_w.data_ptr(),
                  self.lm_b.data_ptr() if self.lm_b is not None else None, self.shape.hidden, self.shape.vocab, self.logits.data_ptr(), st)

    @staticmethod
    def _set_splits(phases, n: int):
        for kind, a in phases:
            if kind == "attn":
                a.nsplit = int(n)
            elif kind == "attn_oproj":                            # n > 1: the long-cache launch (1024 threads; the prefetch share inside the workgroups)
                a.threads = 1024 if n > 1 else 256

    def _variants(self):
        """[(phases, attention splits)]: what runs below / from LONG_FROM positions on (5 launches: the split attention launch)."""
        if self.launches == 4:
            if self.long_from <= 0:
                return [(self.phases, 2)]
            return [(self.phases, 1)] + ([(self.phases, 2)] if self.cache_len > self.long_from else [])
        v = [(self.phases, self.attn_splits)]
        if self.auto_splits and self.cache_len > self.LONG_FROM:
            v.append((self.phases, self.LONG_SPLITS))
        return v

    def _long_threshold(self) -> int:
        return self.long_from if self.launches == 4 else self.LONG_FROM

    def _variant_at(self, pos: int) -> int:
        return 1 if len(self._variants()) > 1 and pos >= self._long_threshold() else 0

    def capture(self):
        """Record one decode step (incl. the position increment) as a hipGraph; replay it with step().  Where a second variant exists
        (5 launches: the split attention from LONG_FROM cached positions on) a second graph is recorded; step() picks by position.
        Quantizers changed since the engine was built (recalibration, scale.copy_) are picked up here, in reset() and in prefill()."""
        if self._keep.stale():
            self._lower()
        tok0, pos0, hp0 = self.tok.clone(), self.pos.clone(), self._host_pos
        graphs = []
        for phases, splits in self._variants():
            self._set_splits(phases, splits)
            self.attn_ticket.zero_()
            with torch.cuda.device(self.dev):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._launch(phases)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch(phases)
                    self.pos.add_(1)
            graphs.append(g)
            self.tok.copy_(tok0); self.pos.copy_(pos0)
        self._host_pos = hp0
        self.graph, self.graph_long = graphs[0], (graphs[1] if len(graphs) > 1 else None)
        return self

    def set_position(self, pos: int):
        """Continue from a cache that already holds `pos` positions (benchmarks; prefill() and reset() call this)."""
        assert 0 <= int(pos) <= self.cache_len
        self.pos.fill_(int(pos))
        self._host_pos = int(pos)

    def cached_values(self, li: int, n: Optional[int] = None) -> torch.Tensor:
        """Layer li's cached values as [kv_heads, n positions, head_dim] int8 indices (index - 128), whatever the engine's layout."""
        n = self._host_pos if n is None else int(n)
        c = self.v_cache[li]
        if not self.v_transposed:
            return c[:, :n]
        s = self.shape
        return c.permute(0, 1, 3, 2).reshape(s.kv_heads, self.cache_len, s.head_dim)[:, :n]      # [kv, chunk, 16, dim] -> [kv, position, dim]

    def load_cached_values(self, li: int, values: torch.Tensor):
        """values [kv_heads, n, head_dim] int8 -> positions 0 .. n - 1 of layer li's value cache."""
        n = values.shape[1]
        if self.v_transposed:
            s, full = self.shape, (n + 15) // 16
            pad = torch.zeros(s.kv_heads, full * 16, s.head_dim, dtype=torch.int8, device=self.dev)
            pad[:, :n] = values
            old = self.v_cache[li][:, :full].permute(0, 1, 3, 2).reshape(s.kv_heads, full * 16, s.head_dim)
            pad[:, n:] = old[:, n:]                                  # (positions behind n in the last chunk keep what they held)
            self.v_cache[li][:, :full] = pad.view(s.kv_heads, full, 16, s.head_dim).permute(0, 1, 3, 2)
        else:
            self.v_cache[li][:, :n] = values

    def fill_cache_random(self, n: int, seed: int = 0):
        """Benchmark helper: n positions of random cached indices (the same logical content for both cache layouts)."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        s = self.shape
        rnd = lambda: torch.randint(-128, 128, (s.kv_heads, n, s.head_dim), generator=g, device=self.dev, dtype=torch.int8)      # noqa: E731
        for c in self.k_cache:
            c[:, :n] = rnd()
        for li in range(len(self.v_cache)):
            self.load_cached_values(li, rnd())
        self.set_position(n)

    def reset(self):
        self._sync_grids()
        self.set_position(0)
        self.attn_ticket.zero_()
        for c in self.k_cache + self.v_cache:
            c.zero_()

    @torch.no_grad()
    def step(self, token: Optional[int] = None) -> torch.Tensor:
        """One token in, logits [vocab] out (device tensor, overwritten by the next step); the position advances by one.
        token None: use the token already sitting in self.tok (e.g. written by a device-side argmax)."""
        if self._host_pos >= self.cache_len:            # the kernels also refuse (they do nothing past the cache); fail loudly here
            raise RuntimeError(f"DecodeEngine.step: the KV cache is full ({self.cache_len} positions); reset() or build a longer cache")
        if token is not None:
            self.tok.fill_(int(token))
        if self.graph is not None:
            (self.graph_long if self.graph_long is not None and self._host_pos >= self._long_threshold() else self.graph).replay()
        else:
            phases, splits = self._variants()[self._variant_at(self._host_pos)]
            self._set_splits(phases, splits)
            with torch.cuda.device(self.dev):
                self._launch(phases)
            self.pos.add_(1)
        self._host_pos += 1
        return self.logits

    @torch.no_grad()
    def prefill(self, context_ids) -> torch.Tensor:
        """Context encoding in ONE forward over the whole context (sim_model.py:176-193) instead of len(context) steps: the module
        graph's prefill (with llama.fuse_decoder_layer: 9 launches per layer) runs with a KV cache attached, the cached keys /
        values are put on their QMatMul input grids (what the step kernels keep in the cache) and the position is set behind the
        context.  Returns the logits of the last context position (self.logits)."""
        ids = torch.as_tensor([int(t) for t in context_ids], dtype=torch.long, device=self.dev).view(1, -1)
        S = ids.shape[1]
        assert 0 < S <= self.cache_len
        self._sync_grids()
        raw = self.model.new_cache(1, S, device=self.dev)
        logits = self.model(ids, cache=raw, last_logits_only=True)      # [1, 1, vocab]: only the last position feeds the first new token
        for li, layer in enumerate(self.model.layers):
            att = layer.self_attn
            self.k_cache[li][:, :S] = att.qk_bmm.input2_quantizer.quantize_to_int(raw[li][0][0].contiguous())[0]
            self.load_cached_values(li, att.pv_bmm.input2_quantizer.quantize_to_int(raw[li][1][0].contiguous())[0])
        self.set_position(S)
        self.logits.copy_(logits[0, -1])
        return self.logits

    @torch.no_grad()
    def generate(self, context_ids, max_new_tokens: int, eos_token_id=None, prefill: bool = True, do_sample: bool = False,
                 temperature: float = 0.5, generator: Optional[torch.Generator] = None):
        """SimModel.generate (mobilellm/model/sim_model.py:160-221): context encoding in one prefill forward (prefill=False: token by
        token through the step kernels), then per new token: next = argmax(logits) or, with do_sample, multinomial(softmax(logits /
        temperature)) (:198-201) -- on the device, into self.tok --, append it, stop if it is an EOS (:202-204), else run the step.
        The host reads one token id per step only to test for EOS and to return the ids."""
        ids = [int(t) for t in context_ids]
        assert len(ids) + max_new_tokens <= self.cache_len
        self.reset()
        if prefill and len(ids) > 1:
            self.prefill(ids)
        else:
            for t in ids:
                self.step(t)
        out = list(ids)
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        for _ in range(max_new_tokens):
            if do_sample:
                probs = torch.softmax(self.logits / temperature, dim=-1)
                self.tok.copy_(torch.multinomial(probs, num_samples=1, generator=generator))
            else:
                torch.argmax(self.logits, dim=-1, keepdim=True, out=self.tok)
            nxt = int(self.tok.item())
            out.append(nxt)
            if nxt in eos:
                break
            if self._host_pos >= self.cache_len:
                break
            self.step()
        return out
