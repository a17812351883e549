"""Per-kernel breakdown of bench.py's e2equant inner step (torch.profiler, device time): where the 14.7 ms go.
    python tools/train_prof.py [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_variants as bench

def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda:0")
    # reuse bench_train_step's construction by running it once (it returns timings only), then re-create the step under the profiler
    import types
    captured = {}
    orig = time.perf_counter
    r = bench.bench_train_step(dev, S)
    print("bench_train_step:", r)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        r = bench.bench_train_step(dev, S)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    rows = sorted(((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.count, e.key) for e in ev if (getattr(e, "device_time_total", 0) or 0) > 0 and e.device_type.name != "CPU"), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"device kernels total {tot / 1e3:.2f} ms over the whole bench_train_step call (6 steps + setup)")
    for t, n, k in rows[:45]:
        print(f"{t / 1e3:9.3f} ms  n={n:5d}  {k[:150]}")

main()
