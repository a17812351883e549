# mq_qmatmul at the attention shapes across library builds: r06_qmm_ab.sh <tag> ...   (prod = the tree's build)
cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  if [ "$tag" != "prod" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== $tag"; python tools/bench_qmatmul.py 2>&1 | grep "S=2048" | cut -c1-80
done
