#!/bin/bash
cat > /tmp/ab.py <<'PY'
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, numpy as np
import mobilequant_amd._lib as L
from mobilequant_amd import ops
from bench_fr128 import timed
dev = torch.device("cuda:0")
def grid(lo, hi, bits):
    n = float(2 ** bits - 1); sc = (hi - lo) / n
    return (torch.tensor([sc], device=dev), torch.tensor([round(-lo / sc)], device=dev, dtype=torch.float32), 0.0, n)
grids = dict(qk_a=grid(-6.0, 6.0, 8), qk_b=grid(-6.0, 6.0, 8), qk_out=grid(-60.0, 60.0, 16), pv_a=grid(0.0, 1.0, 16), pv_b=grid(-4.5, 4.5, 8), pv_out=grid(-2.0, 2.0, 8))
for S, H, KV in ((2048, 32, 4), (2048, 32, 32), (2048, 32, 8), (512, 32, 4), (4096, 32, 4)):
    q, k, v = torch.randn(S, H * 64, device=dev), torch.randn(S, KV * 64, device=dev), torch.randn(S, KV * 64, device=dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32, device=dev) / 64))
    ang = torch.outer(torch.arange(S, dtype=torch.float32, device=dev), inv); ang = torch.cat((ang, ang), -1)
    cos, sin = ang.cos(), ang.sin()
    img = torch.empty(S, H * 64, dtype=torch.int8, device=dev); rs = torch.empty(S, dtype=torch.int32, device=dev)
    res, outs = [], []
    for mode in (1, 2):
        L.load().mq_attention_set_cache(mode)
        res.append(timed(lambda: ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=False)))
        outs.append(img.clone())
    L.load().mq_attention_set_cache(0)
    print(f"S={S} heads {H}/{KV}: small cache {res[0]:.1f} us | deep cache {res[1]:.1f} us (prep + core, graph) identical={bool(torch.equal(outs[0], outs[1]))}", flush=True)
PY
python /tmp/ab.py 2>&1 | grep -v amdgpu
