"""mq_quantize_tiled's staged kernel at the headline activation [2048, 2048] fp32 -> fragment-blocked int8 + row sums: eight rows per
1024-thread workgroup against four per 512 (mq_quantize_tiled_set_rows), identical images, time per launch in one hipGraph."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mobilequant_amd import ops
import mobilequant_amd._lib as L
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
sc, of = torch.tensor([0.031], device=dev), torch.tensor([131.0], device=dev)
outs = {}
for knob in (8, 4, 8, 4):
    L.load().mq_quantize_tiled_set_rows(knob)
    for rows, cols in ((2048, 2048), (2048, 1024), (2048, 4096)):
        x = torch.randn(rows, cols, device=dev) * 2.0
        t = timeit(lambda: ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128))
        print(f"rows/workgroup {knob}: quantize_tiled {rows}x{cols}: {t:.2f} us ({rows * cols * 5 / t / 1e6:.2f} TB/s)")
x = torch.randn(2048, 2048, device=dev) * 2.0
for knob in (8, 4):
    L.load().mq_quantize_tiled_set_rows(knob)
    outs[knob] = ops.quantize_tiled(x, sc, of, 0.0, 255.0, 128); torch.cuda.synchronize()
L.load().mq_quantize_tiled_set_rows(0)
print("identical image / row sums:", bool(torch.equal(outs[8][0], outs[4][0])), bool(torch.equal(outs[8][1], outs[4][1])))
