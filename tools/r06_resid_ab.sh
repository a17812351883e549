# o_proj / w2 (fr128r) and the fused layer across library builds: r06_resid_ab.sh <tag> ...   (prod = the tree's build)
cd $GRAFT_REPO_ROOT
cat > /tmp/rab.py <<'PY'
import torch, bench, bench_variants
from mobilequant_amd import ops
dev = torch.device("cuda:0")
with torch.no_grad():
    r = bench_variants.bench_layer_full(dev, modes=("fused",))
    print("layer", {k: v for k, v in r.items() if "us" in k})
PY
for tag in "$@"; do
  if [ "$tag" != "prod" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== $tag"; PYTHONPATH=$GRAFT_REPO_ROOT python /tmp/rab.py 2>&1 | grep -v amdgpu.ids | tail -2
  PYTHONPATH=$GRAFT_REPO_ROOT python tools/bench_fr128.py 2>&1 | grep -v amdgpu.ids | tail -6
done
