cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
tools/atomic_probe > gpurun_out/r06/atomic_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_round5.py -m gpu -q -x -s -k "fuzz or perplexity" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06/fuzz_ppl.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/decode_base.log
import torch, bench
dev = torch.device("cuda:0")
with torch.no_grad():
    r = bench.bench_decode_full(dev, wbits=8, cache_len=2176, also_contexts=(256, 1024, 2048))
    print({k: r[k] for k in r if "tok" in k or "ms" in k or "context" in k})
PY
cat gpurun_out/r06/atomic_probe.log; tail -5 gpurun_out/r06/fuzz_ppl.log; cat gpurun_out/r06/decode_base.log
