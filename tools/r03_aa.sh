#!/bin/bash
# what-if builds of the fr128r kernel (results are wrong; timing only): no in-loop A loads / no W LDS-DMA / no W fragment reads
for tag in "" noa now nord; do
  if [ -n "$tag" ]; then export MQ_LIB_PATH=$GRAFT_REPO_ROOT/mobilequant_amd/lib/$tag/libmobilequant_amd.so; fi
  echo "== ${tag:-production}"; timeout 200 python tools/bench_fr128.py 2>&1 | grep "o_proj\|^w2"
done
