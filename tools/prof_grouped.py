import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
import mobilequant_amd as mq
dev = torch.device("cuda:0")
M, K, N = 2048, 2048, 5632
x = torch.randn(1, M, K, device=dev) * 1.3
lin = torch.nn.Linear(K, N, bias=False).to(dev)
q = mq.QLinear.from_float(lin, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=4, is_per_channel=True, group_size=128), mq.QuantConfig(bitwidth=8)).requires_grad_(False)
q.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    q(x); q(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): q(x)
        torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]:
    print(f"{e.device_time_total / 5:9.1f} us/call  n={e.count // 5}  {e.key[:110]}")
