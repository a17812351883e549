#!/bin/bash
cat > /tmp/nb.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, mobilequant_amd as mq
from mobilequant_amd.quantization.fp_ops import HFRMSNorm
sys.path.insert(0, "/root/repo/tools")
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
fp = HFRMSNorm(2048, eps=1e-5).to(dev)
norm = mq.QRMSNorm.from_float(fp, a16, a16, a8).requires_grad_(False)
norm.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
x = torch.randn(1, 2048, 2048, device=dev)
with torch.no_grad():
    print("tiled %.2f us | rowmajor %.2f us | full forward %.2f us" % (timeit(lambda: norm.forward_images(x, "tiled")), timeit(lambda: norm.forward_images(x, "rowmajor")), timeit(lambda: norm(x))))
PY
for P in 0 8192 16384 20480 24576 32768 40960 65536; do
echo "pad $P: $(MQ_NORM_PAD_LDS=$P timeout 120 python /tmp/nb.py 2>&1 | grep -v amdgpu | tail -1)"
done
