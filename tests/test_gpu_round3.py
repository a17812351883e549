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
