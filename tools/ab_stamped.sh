#!/bin/bash
# same-box A/B of stamped builds: tags given as arguments (lib/<tag>/), alternated twice
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stamped; mkdir -p $O; cd $R; rm -f $O/ab.log
for rep in 1 2; do for t in "$@"; do
  echo "== $t" >> $O/ab.log
  MQ_LIB_PATH=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so HOLE_ONLY=xcd HOLE_BRIEF=1 timeout 300 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids >> $O/ab.log
done; done
cat $O/ab.log
