import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mobilequant_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=40):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[3]
import mobilequant_amd._lib as L
for rows_per_wg in (8, 4, 8, 4):
  L.load().mq_norm_tiled_set_rows(rows_per_wg)
  print("rows per workgroup", rows_per_wg)
  for rows, cols, ln in ((2048, 2048, False), (2048, 2048, True), (2048, 2560, True)):
      x = torch.randn(rows, cols, device=dev) * 1.7
      w = 1.0 + 0.1 * torch.randn(cols, device=dev)
      b = 0.1 * torch.randn(cols, device=dev) if ln else None
      gi = (torch.tensor([10.0 / 65535], device=dev), torch.tensor([32768.0], device=dev), 0.0, 65535.0)
      go = (torch.tensor([8.0 / 255], device=dev), torch.tensor([128.0], device=dev), 0.0, 255.0)
      t = timeit(lambda: ops.rmsnorm_quant(x, w, b, 1e-5, gi, go, emit_int8=True, layernorm=ln, emit_tiled=True, want_y=False, emit_rowmajor=False))
      print(f"norm_tiled8 {rows}x{cols} ln={ln}: {t:.2f} us  ({(rows*cols*5)/t/1e6:.2f} TB/s)")

# identical images
x = torch.randn(2048, 2048, device=dev) * 1.7; w = 1.0 + 0.1 * torch.randn(2048, device=dev)
gi = (torch.tensor([10.0 / 65535], device=dev), torch.tensor([32768.0], device=dev), 0.0, 65535.0)
go = (torch.tensor([8.0 / 255], device=dev), torch.tensor([128.0], device=dev), 0.0, 255.0)
outs = []
for r in (8, 4):
    L.load().mq_norm_tiled_set_rows(r)
    o = ops.rmsnorm_quant(x, w, None, 1e-5, gi, go, emit_int8=True, layernorm=False, emit_tiled=True, want_y=False, emit_rowmajor=False)
    torch.cuda.synchronize(); outs.append(o)
L.load().mq_norm_tiled_set_rows(0)
print("identical image / row sums:", bool(torch.equal(outs[0][4], outs[1][4])), bool(torch.equal(outs[0][2], outs[1][2])))
