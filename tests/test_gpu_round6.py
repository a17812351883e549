"""Round 6 (GPU): the four-launch decode step (csrc/mq_decode.hip: PAIR_ROPE epilogue, decode_attention_oproj_kernel, OPRE prologue).

The new launches evaluate the SAME expressions as the five-launch chain of rounds 2-5 -- RoPE and the QMatMul input quantizers one
launch earlier, o_proj's contraction as exact integer partial sums inside the attention launch, its epilogue one launch later -- so the
bar is not a tolerance: logits, KV caches and every intermediate the two chains share must be IDENTICAL bit for bit, at every position,
for every leaf graph of BASELINE.json's configs[1..3] (llama / StableLM-2 / Gemma shapes, W8 and packed W4).  The five-launch chain
itself is pinned to the reference's real model by tests/test_gpu_round2.py / round3 (decode_case*.npz, generate_case.npz), which now
run the four-launch engine by default."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


FAMILIES = {
    # name: (LlamaShape kwargs, weight bits, per-channel weights)
    "llama_gqa": (dict(hidden=512, layers=2, heads=8, kv_heads=2, head_dim=64, ffn=1024, vocab=128, max_pos=160), 8, False),
    "llama_gqa_w4": (dict(hidden=512, layers=2, heads=8, kv_heads=2, head_dim=64, ffn=1024, vocab=128, max_pos=160), 4, True),
    "stablelm": (dict(hidden=256, layers=2, heads=4, kv_heads=4, head_dim=64, ffn=512, vocab=96, max_pos=160, norm="layernorm", qkv_bias=True,
                      rotary_pct=0.25), 8, True),
    "gemma_mqa": (dict(hidden=512, layers=2, heads=2, kv_heads=1, head_dim=256, ffn=1024, vocab=128, max_pos=160, hidden_act="gelu",
                       embed_scale=True, eps=1e-6), 4, True),
    "head_dim_32": (dict(hidden=256, layers=2, heads=8, kv_heads=2, head_dim=32, ffn=512, vocab=64, max_pos=160), 8, False),
    "head_dim_128_mha": (dict(hidden=256, layers=1, heads=2, kv_heads=2, head_dim=128, ffn=512, vocab=64, max_pos=160), 8, False),
}


def _model(dev, name):
    import mobilequant_amd as mq
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    kw, wbits, wpc = FAMILIES[name]
    shape = LlamaShape(**kw)
    m = LlamaForCausalLM(shape)
    m.reset_parameters(seed=11, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, shape.vocab, (1, 48), generator=g)
    act = get_act_range(m, [ids, torch.randint(0, shape.vocab, (1, 48), generator=g)])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=wbits, is_per_channel=wpc), a8)
    for n, mod in m.named_modules():                          # ptq/mobilequant.py:175-201
        if isinstance(mod, mq.QLinear):
            if "w2" in n:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in n:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, (mq.QRMSNorm, mq.QLayerNorm)):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
        elif isinstance(mod, mq.QMatMul):
            if "qk_bmm" in n:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in n:
                mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(m, act, "buffer")
    return m, ids[0]


@pytest.mark.parametrize("name", list(FAMILIES))
def test_four_launch_step_is_the_five_launch_step_bit_for_bit(dev, name):
    from mobilequant_amd.decode import DecodeEngine
    m, ids = _model(dev, name)
    e4 = DecodeEngine(m, cache_len=160)
    e5 = DecodeEngine(m, cache_len=160, launches=5, attn_splits=1)
    e4l = DecodeEngine(m, cache_len=160, long_from=0)             # the long-cache (1024-thread) attention + o_proj launch at every position
    assert e4.launches == 4 and e5.launches == 5, "the four-launch kernels must serve this geometry"
    assert len(e4.phases) == 4 * len(m.layers) and len(e5.phases) == 5 * len(m.layers)
    if FAMILIES[name][1] == 4:
        assert all(p[1].w4 == 1 for p in e4.phases if p[0] == "gemv")
    for pos, t in enumerate(ids.tolist()):
        a = e4.step(t).clone()
        b = e5.step(t).clone()
        c = e4l.step(t).clone()
        assert torch.equal(a, b), (name, pos, float((a - b).abs().max()))
        assert torch.equal(c, b), (name, "1024 threads", pos, float((c - b).abs().max()))
    for li in range(len(m.layers)):                            # the RoPE epilogue's cache append == the attention launch's
        n = len(ids)
        assert e4.v_transposed and not e5.v_transposed            # [kv, dim, position] against [kv, position, dim]: compare the logical content
        assert torch.equal(e4.k_cache[li], e5.k_cache[li]) and torch.equal(e4.cached_values(li, n), e5.cached_values(li, n)), (name, li)
    assert torch.equal(e4.x, e5.x)
    # the captured graph replays the same numbers; a second sequence after reset() too (o_proj's accumulators are cleared per step)
    e4.reset()
    e4.capture()
    e5.reset()
    for pos, t in enumerate(ids.tolist()[:20]):
        a = e4.step(t).clone()
        b = e5.step(t).clone()
        assert torch.equal(a, b), (name, "graph", pos)
    # and it still IS the module graph's forward (the budget of test_decode_engine_generic_head_dim_matches_module_graph)
    with torch.no_grad():
        want = m(ids[None, :20].to(dev))[0].cpu().numpy()
    e4.reset()
    got = np.stack([e4.step(int(t)).cpu().numpy().copy() for t in ids[:20]])
    span = float(np.ptp(want))
    d = np.abs(got - want)
    w4 = FAMILIES[name][1] == 4          # (4-bit GeGLU weights on a random model: a flipped nibble-grid index moves a logit further; observed median 0.0015)
    assert d.max() <= 0.05 * span and np.median(d) <= (0.003 if w4 else 0.001) * span and (d <= 0.01 * span).mean() >= (0.9 if w4 else 0.97), \
        (name, d.max() / span, np.median(d) / span, (d <= 0.01 * span).mean())


def test_attention_oproj_launch_against_the_old_launch_pair_and_an_integer_matmul(dev):
    """mq_decode_attention_oproj on its own: the heads' int8 outputs (out_q) are those of mq_decode_attention at the same position and
    cache, and the accumulators hold o_proj's exact integer sums  sum_k w[n, k] a8[k] - w_zp[n] sum_k a8[k]  of that image
    (int64 numpy matmul), after being cleared by the q | k | v launch -- at several positions, incl. 0 and a cache block boundary."""
    from mobilequant_amd import _lib
    from mobilequant_amd.decode import DecodeEngine
    m, ids = _model(dev, "llama_gqa")
    e4 = DecodeEngine(m, cache_len=160, prefetch=0.0)
    e5 = DecodeEngine(m, cache_len=160, launches=5, attn_splits=1, prefetch=0.0)
    st = torch.cuda.current_stream().cuda_stream
    s = m.shape
    out_q = torch.zeros(s.heads * s.head_dim, dtype=torch.int8, device=dev)
    at = e4.phases[1][1]
    at.out_q = out_q.data_ptr()
    o_w, op = e4.oproj_images[0]                                            # [heads][N][D] int8, the o_proj _Linear
    w_nk = o_w.permute(1, 0, 2).reshape(s.hidden, s.heads * s.head_dim).cpu().numpy().astype(np.int64)
    op_zp = op.w_zp.cpu().numpy().astype(np.int64)
    for pos in (0, 1, 63, 64, 65, 130):
        for eng in (e4, e5):
            eng.fill_cache_random(pos, seed=pos)
            eng.x.copy_(torch.randn(s.hidden, generator=torch.Generator().manual_seed(pos)).to(dev) * 2)
        e4.o_acc.fill_(12345)                                                # must be cleared by the first launch
        e4.rope_row.copy_(torch.cat([e4.cos[pos], e4.sin[pos]]))             # (what mq_decode_embed stages at the start of a token)
        _lib.call("mq_decode_gemv", ctypes.byref(e4.phases[0][1]), st)
        _lib.call("mq_decode_attention_oproj", ctypes.byref(at), st)
        _lib.call("mq_decode_gemv", ctypes.byref(e5.phases[0][1]), st)
        _lib.call("mq_decode_attention", ctypes.byref(e5.phases[1][1]), st)
        torch.cuda.synchronize()
        assert torch.equal(out_q, e5.attn_q), pos
        assert torch.equal(e4.k_cache[0], e5.k_cache[0]) and torch.equal(e4.cached_values(0, pos + 1), e5.cached_values(0, pos + 1)), pos
        a8 = out_q.cpu().numpy().astype(np.int64)
        want = w_nk @ a8
        acc = e4.o_acc.cpu().numpy().astype(np.int64)
        assert np.array_equal(acc, want - op_zp * a8.sum()), pos


def test_four_launch_step_on_long_caches_and_block_boundaries(dev):
    """The request schedule of the attention + o_proj launch changes with the position (256 positions requested before *pos is known, 512
    before the scores, later batches inside the sweeps; 16-position chunks of the transposed value cache): at every boundary the logits
    are the five-launch chain's, bit for bit."""
    import dataclasses
    from mobilequant_amd import llama
    from mobilequant_amd.decode import DecodeEngine
    m, ids = _model(dev, "llama_gqa")
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=1600))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    e4 = DecodeEngine(m, cache_len=1600, long_from=10 ** 6)       # the 256-thread launch at every position
    e4l = DecodeEngine(m, cache_len=1600, long_from=0)            # the 1024-thread launch at every position
    e5 = DecodeEngine(m, cache_len=1600, launches=5, attn_splits=1)
    assert e4.launches == 4 and e4l.launches == 4
    for start in (14, 254, 510, 766, 1022, 1278, 1534):
        for eng in (e4, e4l, e5):
            eng.fill_cache_random(start, seed=start)
        for t in (5, 17, 40, 3):                          # the steps cross position start + 2 = a multiple of 16 / 256 / 512 / 1024
            a = e4.step(t).clone()
            c = e4l.step(t).clone()
            b = e5.step(t).clone()
            assert torch.equal(a, b), (start, e4._host_pos, float((a - b).abs().max()))
            assert torch.equal(c, b), (start, "1024 threads", e4._host_pos, float((c - b).abs().max()))
        assert torch.equal(e4.cached_values(0), e5.cached_values(0)) and torch.equal(e4.k_cache[0], e5.k_cache[0])
        assert torch.equal(e4l.cached_values(0), e5.cached_values(0)) and torch.equal(e4l.k_cache[0], e5.k_cache[0])
    # the default engine: two captured graphs, the 1024-thread one from LONG4_FROM cached positions on
    e4 = DecodeEngine(m, cache_len=1600)
    assert e4._long_threshold() == DecodeEngine.LONG4_FROM
    for eng in (e4, e5):
        eng.fill_cache_random(DecodeEngine.LONG4_FROM - 2, seed=5)
        eng.capture()
    assert e4.graph_long is not None
    for t in (5, 17, 40, 3, 90):
        a = e4.step(t).clone()
        b = e5.step(t).clone()
        assert torch.equal(a, b), ("graphs", e4._host_pos, float((a - b).abs().max()))
    # the captured graph across the 512-position batch boundary
    for eng in (e4, e5):
        eng.fill_cache_random(510, seed=9)
        eng.capture()
    for t in (5, 17, 40, 3, 90):
        a = e4.step(t).clone()
        b = e5.step(t).clone()
        assert torch.equal(a, b), ("graph", e4._host_pos, float((a - b).abs().max()))


def test_a_geometry_the_four_launch_kernels_do_not_serve_falls_back_to_five(dev):
    from mobilequant_amd.decode import DecodeEngine
    from mobilequant_amd.llama import LlamaShape
    assert DecodeEngine._oproj_geometry(LlamaShape.tinyllama(), 64) == (8, 1)
    assert DecodeEngine._oproj_geometry(LlamaShape.gemma_2b(), 256) == (32, 2)
    assert DecodeEngine._oproj_geometry(LlamaShape.stablelm_2_1_6b(), 16) == (8, 1)
    assert DecodeEngine._oproj_geometry(LlamaShape(hidden=8192, heads=64, kv_heads=8, head_dim=128), 128) is None      # K > 4096: OPRE prologue


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 5632), (2048, 2048, 2048)])
def test_residual_gemm_family_against_a_numpy_oracle_every_output(dev, M, N, K):
    """the generated residual kernels (`fr128r` / `fr128r8`: w2 and o_proj of the fused layer, x + Q16(linear) in
    fp32) reached the oracle only through "== the C++ kernel" tests.  Here, at the BASELINE shapes (2048, 2048 <- 5632) and (2048, 2048 <-
    2048), every output is compared with numpy: exact int64 contraction, t = acc - w_zp rowsum + col_term, then the epilogue's documented
    fp32 expression (one conversion, ONE fma on the pre-divided constants alpha / s_o and bias / s_o, rint, + offset, clamp to
    the 16-bit grid, (q - o) s_o, + residual) -- the fma evaluated exactly (float64 product + sum; rational arithmetic where the float64
    sum sits on a rounding boundary)."""
    import test_gpu_round2 as T2
    import test_gpu_round3 as T3
    from mobilequant_amd import ops
    a_q, w_q, a_rs, alpha, w_zp, col_term, b = T3._gemm_operands(dev, M, N, K, 7 * M + N + K, False, True)
    resid = torch.randn(M, N, device=dev)
    so_f, oo_f = np.float32(3.1e-4), np.float32(32768.0)
    so, oo = torch.tensor([float(so_f)], device=dev), torch.tensor([float(oo_f)], device=dev)
    got = ops.int8_linear(T3._to_tiled(a_q), w_q, a_rs, alpha, w_zp, col_term, b, resid=resid, a_tiled_rows=M,
                          out_scale=so, out_offset=oo, out_qmin=0.0, out_qmax=65535.0).cpu().numpy()
    F32 = np.float32
    acc = a_q.cpu().numpy().astype(np.int64) @ w_q.cpu().numpy().astype(np.int64).T
    t = acc - w_zp.cpu().numpy().astype(np.int64)[None, :] * a_rs.cpu().numpy().astype(np.int64)[:, None] + col_term.cpu().numpy().astype(np.int64)[None, :]
    assert np.abs(t).max() < 2 ** 31
    inv = F32(1.0) / so_f
    a_p = (alpha.cpu().numpy().astype(F32) * inv).astype(F32)
    b_p = (b.cpu().numpy().astype(F32) * inv).astype(F32)
    tf = t.astype(F32)                                          # v_cvt_f32_i32: round to nearest even
    v64 = tf.astype(np.float64) * a_p.astype(np.float64)[None, :] + b_p.astype(np.float64)[None, :]
    v = v64.astype(F32)
    frac = np.abs(v64 - np.rint(v64))
    for m, n in zip(*np.nonzero(np.abs(frac - 0.5) < 1e-6)):     # the float64 sum (almost) on an index boundary: decide exactly
        v[m, n] = T2._fma_f32_exact(int(tf[m, n]), a_p[n], b_p[n])
    q = np.clip(np.rint(v) + oo_f, F32(0.0), F32(65535.0)).astype(F32)
    y = ((q - oo_f).astype(F32) * so_f).astype(F32)
    want = (resid.cpu().numpy() + y).astype(F32)
    assert q.min() < 20000 and q.max() > 45000                  # the grid is exercised, not saturated
    neq = got.view(np.uint32) != want.view(np.uint32)
    assert not neq.any(), (int(neq.sum()), float(np.abs(got - want).max()))


@pytest.mark.parametrize("tag", ["w8a8", "w4a8"])
def test_four_launch_chain_at_full_size_and_full_occupancy_bit_for_bit(dev, tag):
    """The same identity at TinyLlama-1.1B's real geometry (22 layers, hidden 2048, 32 / 4 heads, FFN 5632: 256 workgroups = one per CU
    in the attention + o_proj launch, 65 536 atomics per layer), from the captured hipGraphs, over 300 steps that cross the 256-position
    batch boundary: the split-K atomics and every cross-launch hand-off (accumulators cleared by one launch, added to by the next, read by
    the third; the chunk-blocked value cache appended while other workgroups sweep it) under the occupancy and timing the benchmark runs
    at.  The model is the contractive one of the perplexity test (the reference's real HFForCausalLM weights)."""
    from test_gpu_round5 import _stable_model
    from mobilequant_amd.decode import DecodeEngine
    import dataclasses
    from mobilequant_amd import llama
    m, z = _stable_model(dev, tag)
    cos, sin = llama.rope_tables(dataclasses.replace(m.shape, max_pos=320))
    m.cos, m.sin = cos.to(dev), sin.to(dev)
    e4 = DecodeEngine(m, cache_len=320)
    e5 = DecodeEngine(m, cache_len=320, launches=5, attn_splits=1)
    assert e4.launches == 4 and len(e4.phases) == 88
    e4.capture()
    e5.capture()
    ids = np.concatenate([z["ids"], z["ids_more"][0]])[:300].tolist()
    for pos, t in enumerate(ids):
        a = e4.step(int(t))
        b = e5.step(int(t))
        if pos % 10 == 0 or pos > 250:
            assert torch.equal(a, b), (tag, pos, float((a - b).abs().max()))
    assert torch.equal(e4.logits, e5.logits) and torch.equal(e4.x, e5.x)
    for li in (0, 10, 21):
        assert torch.equal(e4.k_cache[li][:, :300], e5.k_cache[li][:, :300]) and torch.equal(e4.cached_values(li, 300), e5.cached_values(li, 300)), li


# ---- mq_qmatmul, round 6: the row-panel kernel (x1 <= 8 bits, K <= 256, several column tiles) and the rotated loop orders ----------------
QMM6_CASES = [
    # lead, M, N, K, x2 K-contiguous, (bits, signed) x1, x2, output
    ((2,), 130, 700, 64, True, (8, False), (8, False), (16, False)),      # panel, one 64-k chunk, three column ranges (4 + 4 + 3 tiles), ragged M and N
    ((1,), 200, 520, 100, False, (8, True), (8, False), (8, False)),      # panel, two chunks, N-contiguous x2, K % 64 != 0
    ((1,), 129, 300, 250, True, (6, False), (8, True), None),             # panel, four chunks, K % 4 != 0 (element loads, plain stores), no output quantizer
    ((3,), 260, 1024, 64, True, (8, False), (8, True), (16, True)),       # panel, four column ranges per row panel, rotated tile order
    ((2,), 150, 64, 1000, False, (16, False), (8, False), (8, False)),    # tile kernel, 16 chunks in a rotated order, the partial chunk in the middle
    ((1,), 300, 192, 520, True, (12, True), (8, False), (16, False)),     # tile kernel, two byte planes, three column tiles, 128-row tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", QMM6_CASES, ids=[f"{c[1]}x{c[2]}x{c[3]}_{'kT' if c[4] else 'kn'}_{c[5][0]}b{c[6][0]}b" for c in QMM6_CASES])
def test_qmatmul_row_panel_and_rotated_orders_against_the_exact_integer_oracle(dev, case):
    """mq_qmatmul after round 6 -- the row-panel kernel (a workgroup quantises 128 rows of x1 once and walks a range of column tiles in
    an order rotated per workgroup, buffer stores) and the tile kernel's K loop started at a different chunk per workgroup -- against
    oracle.qmatmul_exact on EVERY output, bit for bit (integer sums: any order gives the same bits)."""
    import test_gpu_round5 as T
    from mobilequant_amd import ops
    from oracle import mq_oracle as O
    lead, M, N, K, kt, (b1, s1), (b2, s2), bo = case
    rng = np.random.default_rng(7000 * M + N + K)
    if b1 > 8 and not s1:       # probabilities
        a = rng.random(lead + (M, K), dtype=np.float32) ** 4
        a /= a.sum(-1, keepdims=True)
        g1 = T._grid(b1, s1, 0.0, float(a.max()))
    else:
        a = (rng.standard_normal(lead + (M, K), dtype=np.float32) * 1.3 + 0.2).astype(np.float32)
        g1 = T._grid(b1, s1, float(a.min()) * 0.9, float(a.max()) * 0.9)
    b = (rng.standard_normal(lead + (K, N), dtype=np.float32) * 0.8 - 0.1).astype(np.float32)
    g2 = T._grid(b2, s2, float(b.min()) * 0.95, float(b.max()) * 0.95)
    fp = np.matmul(a, b)
    go = None if bo is None else T._grid(bo[0], bo[1], float(np.percentile(fp, 0.5)), float(np.percentile(fp, 99.5)))
    want = O.qmatmul_exact(a, b, g1, g2, go)
    ta = torch.from_numpy(a).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(np.swapaxes(b, -1, -2))).to(dev).transpose(-1, -2) if kt else torch.from_numpy(b).to(dev)
    got = ops.qmatmul(ta, tb, T._dev_grid(g1, dev), T._dev_grid(g2, dev), None if go is None else T._dev_grid(go, dev)).cpu().numpy()
    assert got.shape == want.shape
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), (case, int(bad.sum()), np.argwhere(bad)[:5].tolist(), got[bad][:5].tolist(), want[bad][:5].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["tinyllama", "stablelm_2_1_6b", "gemma_2b"])
def test_calibration_mirrored_slots_give_the_statistics_of_the_plain_hooks(dev, family):
    """Round 6: a tensor hooked under several names (a norm's output = the input of q / k / v, w1's output = the activation's input, pv_bmm's
    output = o_proj's input as values) is reduced once from the second forward pass on; the other slots of a group the graph declares
    (LlamaForCausalLM.calibration_alias_groups) mirror the first one after the first pass has confirmed them bit for bit.  The act_dict
    is the one the plain hooks give; 7 reductions per layer + 1 fewer."""
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import ActRangeCollector, get_act_range
    shape = getattr(llama.LlamaShape, family)(layers=2, max_pos=128, vocab=512)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=3, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    samples = [torch.randint(0, shape.vocab, (1, 128), generator=g).to(dev) for _ in range(4)]
    got = {}
    for mirror in (True, False):
        col = ActRangeCollector(model, per_channel=False)
        col.mirror_declared_aliases = mirror
        col.fuse_layer_statistics = False                  # (the one-pass layer glue has a test of its own below)
        col.attach()
        with torch.no_grad():
            for s in samples:
                model(s)
        col.detach()
        got[mirror] = (col.act_dict(), col.bytes_seen, col.bytes_aliased, len(col._mirror))
    assert got[True][0] == got[False][0]
    assert got[False][2] == 0 and got[False][3] == 0
    assert got[True][3] == 7 * shape.layers + 1, got[True][3]
    assert got[True][2] > 0 and got[True][1] + got[True][2] == got[False][1]
    full = get_act_range(model, samples)                    # (+ the one-pass layer glue: within round-off of the plain hooks)
    assert set(full) == set(got[False][0])
    # a declared group whose members do NOT agree after the first pass stays on the plain hooks
    class Wrong(llama.LlamaForCausalLM):
        def calibration_alias_groups(self):
            return [[("layers.0.input_layernorm", "output"), ("layers.0.mlp.w2", "output")]]
    model.__class__ = Wrong
    try:
        col = ActRangeCollector(model, per_channel=False)
        col.fuse_layer_statistics = False
        col.attach()
        with torch.no_grad():
            for s in samples:
                model(s)
        col.detach()
        assert col._mirror == {} and col.act_dict() == got[False][0]
    finally:
        model.__class__ = llama.LlamaForCausalLM


@pytest.mark.gpu
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("shape", [(3, 50, 2048), (2, 7, 1000), (1, 33, 4100)])
def test_calib_norm_and_gated_passes_against_torch(dev, shape, ln):
    """ops.calib_norm_ / ops.calib_gated_ (mq_calib_norm / mq_calib_gated): values within a few ulp of the torch module chain
    (hf_model.py:183-186 / torch layer_norm; silu / gelu times w3), the four running statistics EXACTLY those of the tensors the kernels
    wrote, the residual sum bit for bit."""
    from mobilequant_amd import ops
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 2 + 0.3).to(dev)
    d = torch.randn(*shape, generator=g).to(dev)
    C = shape[-1]
    mod = (torch.nn.LayerNorm(C, eps=1e-5) if ln else HFRMSNorm(C, eps=1e-6)).to(dev)
    with torch.no_grad():
        mod.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1)
        if ln:
            mod.bias.copy_(torch.randn(C, generator=g) * 0.1)
    for delta in (None, d):
        st = [torch.full((1,), v, device=dev) for v in (float("inf"), float("-inf"), float("inf"), float("-inf"))]
        with torch.no_grad():
            h, y = ops.calib_norm_(x, delta, mod.weight, getattr(mod, "bias", None), mod.eps, ln, *st)
            hw = x if delta is None else x + delta
            yw = mod(hw)
        assert torch.equal(h, hw)
        assert torch.allclose(y, yw, rtol=2e-5, atol=2e-6), float((y - yw).abs().max())
        assert [float(s) for s in st] == [float(h.min()), float(h.max()), float(y.min()), float(y.max())]
    a, b = x, d
    for act, fn in (("silu", torch.nn.functional.silu), ("gelu", torch.nn.functional.gelu)):
        st = [torch.full((1,), float("inf") if k % 2 == 0 else float("-inf"), device=dev) for k in range(8)]
        if a.numel() % 4:
            continue
        p = ops.calib_gated_(a, b, act, st)
        s = fn(a)
        assert torch.allclose(p, s * b, rtol=2e-6, atol=1e-7), float((p - s * b).abs().max())
        got = [float(t) for t in st]
        assert got[0:2] == [float(a.min()), float(a.max())] and got[4:6] == [float(b.min()), float(b.max())]
        assert got[6:8] == [float(p.min()), float(p.max())]
        assert abs(got[2] - float(s.min())) <= 1e-6 * max(1, abs(float(s.min()))) and abs(got[3] - float(s.max())) <= 2e-6 * abs(float(s.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["tinyllama", "stablelm_2_1_6b", "gemma_2b"])
def test_calibration_layer_passes_give_the_act_dict_of_the_plain_hooks(dev, family):
    """Round 6: while ONE collector is attached, llama.DecoderLayer / MLP take the norms (both statistics + the residual add in one pass)
    and act(w1) * w3 (four statistics in one pass) from mq_calib_norm / mq_calib_gated.  Same keys, values within 1e-5 relative of the
    plain hooks (row sums in another order), the logits within fp32 round-off; a second collector on the same model sends both back to
    the plain hooks."""
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import ActRangeCollector
    shape = getattr(llama.LlamaShape, family)(layers=2, max_pos=128, vocab=512)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=3, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    samples = [torch.randint(0, shape.vocab, (1, 128), generator=g).to(dev) for _ in range(3)]
    got, logits = {}, {}
    for fuse in (True, False):
        col = ActRangeCollector(model, per_channel=False)
        col.fuse_layer_statistics = fuse
        col.attach()
        assert (len(col._layers) == 3 * shape.layers + 1) == fuse       # decoder layer (norms), MLP (gated product), attention (RoPE); the model (final norm)
        with torch.no_grad():
            for s in samples:
                logits[fuse] = model(s)
        col.detach()
        got[fuse] = (col.act_dict(), col.bytes_seen, col.bytes_fused)
    assert not any("_mq_calib_layer" in m.__dict__ for m in model.modules())
    assert set(got[True][0]) == set(got[False][0])
    worst = 0.0
    for name, fields in got[False][0].items():
        assert set(fields) == set(got[True][0][name]), name
        for f, (lo, hi) in fields.items():
            a = got[True][0][name][f]
            for u, v in ((a[0], lo), (a[1], hi)):
                worst = max(worst, abs(u - v) / max(abs(v), 1e-3))
    assert worst <= 1e-5, worst
    assert torch.allclose(logits[True], logits[False], rtol=1e-3, atol=1e-3)
    assert got[True][1] < 0.75 * got[False][1] and got[True][2] > got[False][2]
    # two collectors: both on the plain hooks, both complete
    c1 = ActRangeCollector(model, per_channel=False).attach()
    c2 = ActRangeCollector(model, per_channel=False).attach()
    assert not c1._layers and not c2._layers
    with torch.no_grad():
        for s in samples:
            model(s)
    c1.detach(); c2.detach()
    d1, d2 = c1.act_dict(), c2.act_dict()
    assert set(d1) == set(d2) == set(got[False][0])
    for name, fields in got[False][0].items():
        for f, (lo, hi) in fields.items():
            for dd in (d1, d2):
                assert abs(dd[name][f][0] - lo) <= 1e-5 * max(abs(lo), 1e-3) and abs(dd[name][f][1] - hi) <= 1e-5 * max(abs(hi), 1e-3), (name, f)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(2, 37, 32, 4, 64, 64), (1, 50, 8, 1, 256, 256), (1, 29, 32, 32, 64, 16), (2, 21, 6, 2, 80, 32)])
def test_calib_rope_pass_is_apply_rope_bit_for_bit(dev, geom):
    """ops.calib_rope_ (mq_calib_rope): the rotated q / k of llama.apply_rope bit for bit -- full and partial rotary, GQA -- in
    [B, heads, S, D] order, and the four statistics exactly those of the projections' outputs and of the rotated tensors."""
    from mobilequant_amd import ops, llama
    B, S, H, KV, D, rot = geom
    g = torch.Generator().manual_seed(S)
    ql = (torch.randn(B, S, H * D, generator=g) * 1.7).to(dev)
    kl = (torch.randn(B, S, KV * D, generator=g) * 0.6 + 0.2).to(dev)
    shape = llama.LlamaShape(hidden=64, layers=1, heads=H, kv_heads=KV, head_dim=D, ffn=64, vocab=16, max_pos=128, rot_dim=rot) if "rot_dim" in llama.LlamaShape.__dataclass_fields__ else None
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    ang = torch.outer(torch.arange(S, dtype=torch.float32) + 5, inv)
    ang = torch.cat((ang, ang), dim=-1)
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)
    st = [torch.full((1,), float("inf") if k % 2 == 0 else float("-inf"), device=dev) for k in range(8)]
    q, k = ops.calib_rope_(ql, kl, H, KV, D, cos, sin, st)
    qw = llama.apply_rope(ql.view(B, S, H, D).transpose(1, 2), cos, sin)
    kw = llama.apply_rope(kl.view(B, S, KV, D).transpose(1, 2), cos, sin)
    assert q.is_contiguous() and k.is_contiguous() and q.shape == qw.shape and k.shape == kw.shape
    assert torch.equal(q, qw) and torch.equal(k, kw)
    want = [ql.min(), ql.max(), qw.min(), qw.max(), kl.min(), kl.max(), kw.min(), kw.max()]
    assert [float(t) for t in st] == [float(t) for t in want]


@pytest.mark.gpu
@pytest.mark.parametrize("S", [64, 100, 2048])
def test_causal_score_chain_without_a_mask_tensor_and_without_the_masked_stores(dev, S):
    """ops.calib_attention_probs_causal_ (mq_calib_attention_probs_causal): the probabilities and the four statistics of
    calib_attention_probs_ under the explicit causal mask, bit for bit; with store_masked = False the buffer's upper triangles are left
    alone (zeros stay zeros from call to call, stale non-zeros would stay too) and the lower triangles are those of the full pass."""
    from mobilequant_amd import ops
    H = 3 if S < 2048 else 2
    g = torch.Generator().manual_seed(S)
    raws = [(torch.randn(1, H, S, S, generator=g) * 3).to(dev) for _ in range(2)]
    mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
    new = lambda: [torch.full((1,), float("inf") if k % 2 == 0 else float("-inf"), device=dev) for k in range(4)]
    buf = torch.zeros_like(raws[0])
    for raw in raws:
        st_ref, st_c, st_k = new(), new(), new()
        want = ops.calib_attention_probs_(raw.clone(), mask, 8.0, *st_ref)
        full = ops.calib_attention_probs_causal_(raw, torch.empty_like(raw), 8.0, True, *st_c)
        kept = ops.calib_attention_probs_causal_(raw, buf, 8.0, False, *st_k)
        assert torch.equal(full, want) and torch.equal(kept, want) and kept.data_ptr() == buf.data_ptr()
        assert [float(t) for t in st_ref] == [float(t) for t in st_c] == [float(t) for t in st_k]
    junk = torch.full_like(raws[0], 7.0)
    ops.calib_attention_probs_causal_(raws[0], junk, 8.0, False, *new())
    lower = torch.ones(S, S, device=dev).tril().bool()
    assert torch.equal(junk[..., lower], want_lower := ops.calib_attention_probs_(raws[0].clone(), mask, 8.0, *new())[..., lower])
    q = torch.arange(S, device=dev)
    wholly_above = (q[None, :] // 4 * 4) > q[:, None]           # quads that start beyond the diagonal were not stored
    assert bool((junk[..., wholly_above] == 7.0).all())


@pytest.mark.gpu
def test_calibration_with_the_kept_causal_zeros_is_the_calibration_without(dev):
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import ActRangeCollector, get_act_range
    shape = llama.LlamaShape.tinyllama(layers=2, max_pos=128, vocab=512)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=3, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    samples = [torch.randint(0, shape.vocab, (1, n), generator=g).to(dev) for n in (128, 128, 96, 128)]      # a shape change in between
    a = get_act_range(model, samples)
    ActRangeCollector.keep_causal_zeros = False
    try:
        b = get_act_range(model, samples)
    finally:
        ActRangeCollector.keep_causal_zeros = True
    assert a == b


@pytest.mark.gpu
def test_per_channel_calibration_mirrors_the_groups_that_are_one_tensor(dev):
    """Per-channel mode (act ranges per feature, SmoothQuant's absmax): the declared alias groups whose members are ONE tensor are
    mirrored after the first pass as in per-tensor mode; the value-only group (pv_bmm.output / o_proj.input: other channel counts) fails
    the check and keeps its hooks.  Same act_dict and act_scales as without mirroring."""
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import ActRangeCollector
    shape = llama.LlamaShape.tinyllama(layers=2, max_pos=128, vocab=512)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=3, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    samples = [torch.randint(0, shape.vocab, (1, 128), generator=g).to(dev) for _ in range(3)]
    got = {}
    for mirror in (True, False):
        col = ActRangeCollector(model, per_channel=True)
        col.mirror_declared_aliases = mirror
        col.attach()
        with torch.no_grad():
            for s in samples:
                model(s)
        col.detach()
        got[mirror] = (col.act_dict(), col.act_scales(), len(col._mirror), col.bytes_aliased)
    assert got[True][2] == 6 * shape.layers + 1 and got[False][2] == 0 and got[True][3] > 0
    a, b = got[True][0], got[False][0]
    assert a.keys() == b.keys()
    for name in a:
        assert a[name].keys() == b[name].keys()
        for f in a[name]:
            assert torch.equal(a[name][f], b[name][f]), (name, f)
    assert got[True][1].keys() == got[False][1].keys() and all(torch.equal(got[True][1][k], got[False][1][k]) for k in got[True][1])


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(2, 37, 32, 4, 64, 64), (1, 29, 8, 8, 64, 16), (1, 19, 6, 2, 80, 32)])
def test_calib_rope_qkv_pass_is_apply_rope_and_repeat_kv_bit_for_bit(dev, geom):
    from mobilequant_amd import ops, llama
    B, S, H, KV, D, rot = geom
    g = torch.Generator().manual_seed(S)
    ql = (torch.randn(B, S, H * D, generator=g) * 1.7).to(dev)
    kl = (torch.randn(B, S, KV * D, generator=g) * 0.6 + 0.2).to(dev)
    vl = (torch.randn(B, S, KV * D, generator=g) * 2.5 - 0.4).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    ang = torch.outer(torch.arange(S, dtype=torch.float32) + 3, inv)
    ang = torch.cat((ang, ang), dim=-1)
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)
    st = [torch.full((1,), float("inf") if k % 2 == 0 else float("-inf"), device=dev) for k in range(10)]
    q, k, v = ops.calib_rope_qkv_(ql, kl, vl, H, KV, D, cos, sin, st)
    rep = H // KV
    expand = lambda t: t[:, :, None].expand(B, KV, rep, S, D).reshape(B, H, S, D)
    qw = llama.apply_rope(ql.view(B, S, H, D).transpose(1, 2), cos, sin)
    kw = expand(llama.apply_rope(kl.view(B, S, KV, D).transpose(1, 2), cos, sin))
    vw = expand(vl.view(B, S, KV, D).transpose(1, 2))
    assert torch.equal(q, qw) and torch.equal(k, kw) and torch.equal(v, vw)
    want = [ql.min(), ql.max(), qw.min(), qw.max(), kl.min(), kl.max(), kw.min(), kw.max(), vl.min(), vl.max()]
    assert [float(t) for t in st] == [float(t) for t in want]
