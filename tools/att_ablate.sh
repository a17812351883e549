#!/bin/bash
# A/B of tagged builds of the attention kernel (mobilequant_amd/lib/<tag>/, tools: build.build(tag=...)) against the production build on one box
# usage (under gpurun): bash tools/att_ablate.sh tag1 tag2 ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/att_ablate; mkdir -p $OUT; cd $R; : > $OUT/ablate.log
for t in "" "$@"; do
  L=mobilequant_amd/lib/$t/libmobilequant_amd.so
  [ -f $L ] || continue
  echo "== ${t:-production}" >> $OUT/ablate.log
  for i in 1 2; do MQ_LIB_PATH=$L MQ_ATT_F16=1 python tools/prof_attention.py 2>&1 | grep -v amdgpu.ids >> $OUT/ablate.log; done
done
cat $OUT/ablate.log
