"""The fused whole-layer prefill of bench.py (layer_prefill_full, mode "fused" only) for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_variants as bench  # noqa: E402

print(bench.bench_layer_full(torch.device("cuda:0"), modes=tuple(os.environ.get("MQ_LAYER_MODES", "fused").split(",")),
                             wbits=int(os.environ.get("MQ_LAYER_WBITS", "8"))))
