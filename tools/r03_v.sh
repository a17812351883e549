#!/bin/bash
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, bench
print(bench.bench_train_step(torch.device("cuda:0"), S=512))
print(bench.bench_train_step(torch.device("cuda:0"), S=2048))
PY
