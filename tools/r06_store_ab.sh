# headline GEMM with other cache policies on its output stores (generator switch MQ_FR_STORE): step time and GEMM-alone period per build
cd $GRAFT_REPO_ROOT
cat > /tmp/st_ab.py <<'PY'
import torch, bench
from mobilequant_amd._lib import MQ_U8
dev = torch.device("cuda:0")
with torch.no_grad():
    step = bench.Step(dev, MQ_U8, seed=0)
    sec = bench.run_steps(step, 400, 20, 1)
    g = [round(bench.event_time(step.gemm, 50) * 1e6, 2) for _ in range(3)]
    sec2 = bench.run_steps(step, 400, 20, 1)
    print("step_us", round(sec * 1e6, 2), round(sec2 * 1e6, 2), "gemm_alone_us", g)
PY
for tag in "$@"; do
  if [ "$tag" != "prod" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== $tag"; PYTHONPATH=$GRAFT_REPO_ROOT python /tmp/st_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
done
