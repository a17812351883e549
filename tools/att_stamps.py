"""Where a wave of the f16 attention kernel spends its cycles (-DMQ_ATT_STAMPS build of mq_attention.hip, lib/attst/): s_memtime
differences accumulated per wave over q preparation, the waits / LDS + MFMA issue / quantizer chain of sweep 1, the waits and the rest
of sweep 2, the epilogue.      MQ_LIB_PATH=mobilequant_amd/lib/attst/libmobilequant_amd.so python tools/att_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilequant_amd import ops  # noqa: E402

S, H, KV, D = int(os.environ.get("MQ_ATT_S", 2048)), 32, 4, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = torch.randn(S, H * D, device=dev), torch.randn(S, KV * D, device=dev), torch.randn(S, KV * D, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
ang = torch.outer(torch.arange(S, dtype=torch.float32, device=dev), inv)
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos(), ang.sin()


def grid(lo, hi, bits):
    n = float(2 ** bits - 1)
    sc = (hi - lo) / n
    return (torch.tensor([sc], device=dev), torch.tensor([round(-lo / sc)], device=dev, dtype=torch.float32), 0.0, n)


grids = dict(qk_a=grid(-6.0, 6.0, 8), qk_b=grid(-6.0, 6.0, 8), qk_out=grid(-60.0, 60.0, 16), pv_a=grid(0.0, 1.0, 16), pv_b=grid(-4.5, 4.5, 8),
             pv_out=grid(-2.0, 2.0, 8))
for _ in range(3):
    out = ops.attention_quant(q, k, v, cos, sin, H, KV, grids)
torch.cuda.synchronize()
st = out.view(torch.int64).flatten()[: (S // 64) * H * 4 * 11].view(-1, 11).double().cpu()
names = ["q prep", "s1 wait", "s1 lds+mfma issue", "s1 chain", "s2 wait", "s2 rest", "epilogue", "whole wave", "key blocks", "s1 DMA issue", "s1 one ds_read round trip"]
print("waves", st.shape[0], "mean cycles per wave:")
for i, n in enumerate(names):
    print(f"  {n:20s} {st[:, i].mean():10.0f}")
nb = st[:, 8]
print("per key block (sum over waves / sum of blocks): s1 wait %.0f, s1 DMA issue %.0f, one ds_read %.0f, s1 lds+mfma %.0f, s1 chain %.0f, s2 wait %.0f, s2 rest %.0f" % tuple(
    float(st[:, i].sum() / nb.sum()) for i in (1, 9, 10, 2, 3, 4, 5)))
for lo, hi in ((1, 4), (8, 12), (28, 32)):
    m = (nb >= lo) & (nb <= hi)
    print(f"  rows with {lo}-{hi} key blocks: whole {st[m, 7].mean():.0f}, q prep {st[m, 0].mean():.0f}, s1 {st[m, 1:4].sum(1).mean():.0f} (wait {st[m, 1].mean():.0f}), "
          f"s2 {st[m, 4:6].sum(1).mean():.0f} (wait {st[m, 4].mean():.0f}), epilogue {st[m, 6].mean():.0f}")
