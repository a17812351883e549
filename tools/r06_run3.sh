cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06/round6_tests.log
tail -15 gpurun_out/r06/round6_tests.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/decode_4launch.log
import torch, bench
dev = torch.device("cuda:0")
with torch.no_grad():
    for L in (4, 5):
        r = bench.bench_decode_full(dev, wbits=8, cache_len=2176, also_contexts=(256, 512, 1024, 2048), launches=L)
        print(L, {k: r[k] for k in r if "tok" in k or "ms" in k or "context" in k})
PY
cat gpurun_out/r06/decode_4launch.log
bash tools/r06_stamps.sh
