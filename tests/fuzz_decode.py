"""Random-geometry fuzz of the four-launch decode step (test infrastructure; run on an MI355X: python tests/fuzz_decode.py [cases]): for random leaf
graphs (heads / KV heads / head_dim / hidden / FFN / rotary fraction / norm kind / activation / q|k|v bias / weight bits) and random cache
fills, the four-launch engine's logits, key cache and value cache must equal the five-launch engine's BIT FOR BIT over several steps
(tests/test_gpu_round6.py pins six fixed graphs; this walks the space between them).  A geometry the four-launch kernels do not serve must
fall back to five launches (and is then skipped)."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import mobilequant_amd as mq  # noqa: E402
from mobilequant_amd.calibration import get_act_range  # noqa: E402
from mobilequant_amd.decode import DecodeEngine  # noqa: E402
from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(os.environ.get("SEED", "6")))
bad = served = 0
for it in range(n_cases):
    D = int(rng.choice([32, 64, 64, 128, 256]))
    heads = int(rng.choice([h for h in (1, 2, 4, 8, 16) if (h * D) % 256 == 0 and h * D <= 2048]))      # (the five-launch o_proj GEMV wants K % 256 == 0)
    kv = int(rng.choice([h for h in (1, 2, 4, 8, 16) if heads % h == 0]))
    hidden = 256 * int(rng.integers(1, 5))
    ffn = 256 * int(rng.integers(1, 5))
    rot = float(rng.choice([1.0, 1.0, 0.5, 0.25]))
    ln = bool(rng.integers(0, 2))
    wbits = int(rng.choice([8, 8, 4]))
    cache_len = 16 * int(rng.integers(6, 40))
    shape = LlamaShape(hidden=hidden, layers=1, heads=heads, kv_heads=kv, head_dim=D, ffn=ffn, vocab=64, max_pos=cache_len, rotary_pct=rot,
                       norm="layernorm" if ln else "rmsnorm", qkv_bias=bool(rng.integers(0, 2)), hidden_act=str(rng.choice(["silu", "gelu"])))
    try:
        m = LlamaForCausalLM(shape)
    except Exception as e:                                                   # (a geometry the model class itself refuses)
        print("skip model", it, shape, e)
        continue
    m.reset_parameters(seed=100 + it, std=0.08)
    m = m.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(it)
    ids = torch.randint(0, 64, (1, 24), generator=g)
    act = get_act_range(m, [ids, torch.randint(0, 64, (1, 24), generator=g)])
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=wbits, is_per_channel=wbits == 4 or bool(rng.integers(0, 2))), mq.QuantConfig(bitwidth=8))
    for n, mod in m.named_modules():
        if isinstance(mod, mq.QLinear) and ("w2" in n or "o_proj" in n):
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
                mod.input_quantizer.qcfg.bitwidth = int(rng.choice([16, 12, 8]))
    mq.set_scale_and_offset(m, act, "buffer")
    try:
        e4 = DecodeEngine(m, cache_len=cache_len)
        e5 = DecodeEngine(m, cache_len=cache_len, launches=5, attn_splits=1)
    except RuntimeError as e:
        print("skip engine", it, e)
        continue
    if e4.launches != 4:
        print("fallback", it, D, heads, kv, hidden)
        continue
    served += 1
    start = int(rng.integers(0, cache_len - 6))
    for eng in (e4, e5):
        eng.fill_cache_random(start, seed=it)
    ok = True
    for t in rng.integers(0, 64, size=5).tolist():
        a, b = e4.step(int(t)).clone(), e5.step(int(t)).clone()
        ok = ok and torch.equal(a, b)
    n = start + 5
    ok = ok and torch.equal(e4.k_cache[0][:, :n], e5.k_cache[0][:, :n]) and torch.equal(e4.cached_values(0, n), e5.cached_values(0, n))
    if not ok:
        bad += 1
        print("MISMATCH", it, dict(D=D, heads=heads, kv=kv, hidden=shape.hidden, ffn=ffn, rot=rot, ln=ln, wbits=wbits, cache_len=cache_len, start=start), flush=True)
torch.cuda.synchronize()
print("fuzz_decode: cases", n_cases, "served by the four-launch kernels", served, "bad", bad)
sys.exit(1 if bad or served == 0 else 0)
