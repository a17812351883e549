#!/bin/bash
O=gpurun_out/r03z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "quantize or tiled or int8 or qlinear" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
import mobilequant_amd._lib as L
from mobilequant_amd import ops
from bench_fr128 import timed
dev = torch.device("cuda:0")
for rows, cols in ((2048, 2048), (2048, 4096), (4096, 2048)):
    x = torch.randn(rows, cols, device=dev)
    sc, of = torch.tensor([0.031], device=dev), torch.tensor([131.0], device=dev)
    res = []
    for on in (0, 1):
        L.load().mq_quantize_tiled_set_staged(on)
        res.append(timed(lambda: ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128)))
    L.load().mq_quantize_tiled_set_staged(1)
    print(f"quantize_tiled [{rows}, {cols}]: lane-per-fragment {res[0]:.2f} us | staged {res[1]:.2f} us (both allocate outputs)")
PY
