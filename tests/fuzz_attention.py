"""Random-shape fuzz of mq_attention_quant against the oracle (test infrastructure; run by hand on an MI355X: python tests/fuzz_attention.py)."""
import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import mq_oracle as O
from mobilequant_amd import ops
import test_gpu_round2 as T
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
bad = 0
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(n_cases):
    S = int(rng.integers(2, 420))
    heads = int(rng.choice([1, 2, 4, 8]))
    kv = int(rng.choice([h for h in (1, 2, 4, 8) if heads % h == 0]))
    qkb = int(rng.choice([16, 16, 12, 0]))
    pvb = int(rng.choice([8, 8, 16, 0]))
    q, k, v, cos, sin, qk, pv = T._attention_case(S, heads, kv, seed=1000 + it, qk_out_bits=qkb, pv_out_bits=pvb)
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    q, k = q * scale, k / scale   # other grids
    q, k, v, cos, sin, qk, pv = T._attention_case(S, heads, kv, seed=1000 + it, qk_out_bits=qkb, pv_out_bits=pvb)
    want = O.attention_sim(q, k, v, cos, sin, heads, kv, qk, pv)
    grids = dict(qk_a=T._grid_of(qk[0], dev), qk_b=T._grid_of(qk[1], dev), qk_out=T._grid_of(qk[2], dev), pv_a=T._grid_of(pv[0], dev),
                 pv_b=T._grid_of(pv[1], dev), pv_out=T._grid_of(pv[2], dev))
    t = lambda a: torch.from_numpy(a).to(dev)
    got = ops.attention_quant(t(q), t(k), t(v), t(cos), t(sin), heads, kv, grids).cpu().numpy()
    d = np.abs(got - want)
    span = float(want.max() - want.min())
    step = float(pv[2].scale) if pv[2] is not None else 0.0
    ok = np.isfinite(got).all() and d.max() <= max(1.001 * step, 2e-3 * span) and np.median(d) <= 2e-4 * span
    frac = float((d > 0.5 * step).mean()) if step else 0.0
    if not ok or (pvb == 8 and frac > 0.02):
        bad += 1
        print("BAD", it, S, heads, kv, qkb, pvb, d.max(), step, span, frac)
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
