import torch, sys, os
sys.path.insert(0, os.getcwd())
from mobilequant_amd import ops
dev=torch.device("cuda:0")
for shape in ((2048,2048),(5632,2048),(2048,5632)):
    x=torch.randn(*shape,device=dev)
    def fresh():
        mn,mx=ops.minmax_new(1,dev); ops.minmax_tensor_(x,mn,mx)
    mn,mx=ops.minmax_new(1,dev); ops.minmax_tensor_(x,mn,mx)
    def running(): ops.minmax_tensor_(x,mn,mx)
    for name,fn in (("fresh",fresh),("running",running)):
        fn(); torch.cuda.synchronize()
        g=torch.cuda.CUDAGraph()
        s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): fn()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        t=e0.elapsed_time(e1)/20*1e3
        print(shape,name,f"{t:.2f} us  {x.numel()*4/t/1e6:.2f} TB/s")
    assert mn.item()==x.min().item() and mx.item()==x.max().item()
