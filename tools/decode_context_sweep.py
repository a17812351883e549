#!/usr/bin/env python3
"""ms per token of the full TinyLlama W8A8 decode step against the number of cached positions, for fixed attention splits and for
the position-dependent default (DecodeEngine.LONG_FROM / LONG_SPLITS)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

dev = torch.device("cuda:0")
ctxs = [128, 256, 512, 640, 768, 896, 1024, 1536, 1984]
for splits in (1, 2, 4, 8, None):
    r = bench.bench_decode_full(dev, steps=48, cache_len=2048, attn_splits=splits, contexts=ctxs)
    print(f"splits={splits}: " + "  ".join(f"{c}:{v:.4f}" for c, v in r.items()), flush=True)
