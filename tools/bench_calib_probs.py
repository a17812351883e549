#!/usr/bin/env python3
"""mq_calib_attention_probs (calibration-mode score chain, in place) at [32, S, S], S = 2048: with the causal additive mask, without a mask.
Algorithmic bytes: 8 per score (read raw, write probabilities); the mask row is a second, cache-resident read."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mobilequant_amd import ops
dev = torch.device("cuda:0")
H, S = 32, 2048
raw0 = torch.randn(1, H, S, S, device=dev)
mask = torch.full((S, S), float("-inf"), device=dev).triu(1)
st = [torch.zeros(1, device=dev) for _ in range(4)]


def timed(fn, n=5, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


raw = raw0.clone()
for name, m in (("causal mask", mask), ("no mask", None)):
    us = timed(lambda: ops.calib_attention_probs_(raw, m, 8.0, *st))
    print(f"{name}: {us:.1f} us = {8 * raw.numel() / us / 1e6:.2f} TB/s of algorithmic bytes", flush=True)
buf = torch.zeros_like(raw)
for name, sm in (("causal, no mask tensor, all stores", True), ("causal, masked quads not stored (kept zeros)", False)):
    src = raw0.clone()
    us = timed(lambda: ops.calib_attention_probs_causal_(src, buf, 8.0, sm, *st))
    print(f"{name}: {us:.1f} us = {8 * raw.numel() / us / 1e6:.2f} TB/s of the full pass's algorithmic bytes", flush=True)
