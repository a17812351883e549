# A/B of two library builds on one box: the headline step, the GEMM alone (bench.py's event_time) and one fused TinyLlama layer, interleaved
cd $GRAFT_REPO_ROOT
cat > /tmp/lean_ab.py <<'PY'
import torch, bench, bench_variants
from mobilequant_amd._lib import MQ_U8
dev = torch.device("cuda:0")
with torch.no_grad():
    step = bench.Step(dev, MQ_U8, seed=0)
    sec = bench.run_steps(step, 400, 20, 1)
    g = [round(bench.event_time(step.gemm, 50) * 1e6, 2) for _ in range(4)]
    sec2 = bench.run_steps(step, 400, 20, 1)
    print("step_us", round(sec * 1e6, 2), round(sec2 * 1e6, 2), "gemm_alone_us", g)
    if LAYER:
        r = bench_variants.bench_layer_full(dev, modes=("fused",))
        print("layer", {k: v for k, v in r.items() if "us" in k})
PY
sed -i "s/LAYER/${LAYER:-0}/" /tmp/lean_ab.py
for tag in prod lean prod lean prod lean; do
  if [ "$tag" != "prod" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== $tag"; PYTHONPATH=$GRAFT_REPO_ROOT python /tmp/lean_ab.py 2>&1 | grep -v amdgpu.ids | tail -3
done
