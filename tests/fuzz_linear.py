"""Random-shape consistency fuzz of the GEMM family's round-2 additions (run on an MI355X): the residual store and the segmented
output grids must equal the plain launches bit for bit on every tile variant the heuristic picks."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import mobilequant_amd as mq  # noqa: E402
from mobilequant_amd import ops  # noqa: E402
from mobilequant_amd._lib import MQ_F32, MQ_I8, MQ_U8  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
bad = 0
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(n_cases):
    M = int(rng.choice([9, 33, 200, 777, 2048, 4096]))
    K = int(rng.choice([128, 256, 1024, 2048]))
    Ns = [int(rng.choice([4, 36, 64, 128, 256, 1000, 2048])) for _ in range(int(rng.integers(1, 4)))]
    g = torch.Generator(device="cpu").manual_seed(it)
    x = torch.randn(M, K, generator=g).to(dev)
    aq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
    aq.set_scale_offset_from_minmax(float(x.min()), float(x.max()), "buffer", dev)
    a_q, a_rs, a_shift = aq.quantize_to_int(x, MQ_I8, want_row_sum=True)
    parts, singles, grids = [], [], []
    for i, N in enumerate(Ns):
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        bias = (torch.randn(N, generator=g) * 0.1).to(dev) if it % 2 else None
        wq = mq.Quantizer(mq.QuantConfig(bitwidth=8, is_per_channel=bool(i & 1)))
        wq(w)
        w8, colsum, wshift = wq.quantize_to_int(w, MQ_I8, want_row_sum=True, rows=N)
        alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, a_shift, wq.scale.detach(), wq.offset.detach(), wshift, colsum, K)
        y = torch.nn.functional.linear(x, w, bias)
        oq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
        oq.set_scale_offset_from_minmax(float(y.min()) * 0.8, float(y.max()) * 0.8, "buffer", dev)
        kw = dict(out_scale=oq.scale, out_offset=oq.offset, out_qmin=0.0, out_qmax=255.0)
        singles.append(ops.int8_linear(a_q, w8, a_rs, alpha, wzp, ct, bias, out_dtype=MQ_U8, **kw))
        parts.append((w8, alpha, wzp, ct, bias))
        grids.append((oq.scale, oq.offset))
        # residual store (fp32 output; 16-bit grid as o_proj / w2 use, and no output quantizer)
        o16 = mq.Quantizer(mq.QuantConfig(bitwidth=16))
        o16.set_scale_offset_from_minmax(float(y.min()), float(y.max()), "buffer", dev)
        res = torch.randn(M, N, generator=g).to(dev)
        for kw2 in (dict(out_scale=o16.scale, out_offset=o16.offset, out_qmin=float(o16.qmin), out_qmax=float(o16.qmax)), dict()):
            plain = ops.int8_linear(a_q, w8, a_rs, alpha, wzp, ct, bias, out_dtype=MQ_F32, **kw2)
            fused = ops.int8_linear(a_q, w8, a_rs, alpha, wzp, ct, bias, out_dtype=MQ_F32, resid=res, **kw2)
            if not torch.equal(fused, res + plain):
                bad += 1
                print("BAD resid", it, M, K, N, bool(kw2))
    cat = [torch.cat([p[j] for p in parts]) if parts[0][j] is not None else None for j in range(5)]
    got = ops.int8_linear_segmented(a_q, cat[0], a_rs, cat[1], cat[2], cat[3], cat[4], np.cumsum(Ns).tolist(), grids)
    if not torch.equal(got, torch.cat(singles, dim=1)):
        bad += 1
        print("BAD segmented", it, M, K, Ns)
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
