#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
export MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so DVFS_FILLS=gauss DVFS_VARIANTS=11 DVFS_EAGER=1 DVFS_GAP=1
for k in default 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$k" >> $O/gap.log
  if [ $k = default ]; then timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids >> $O/gap.log
  else HIP_FORCE_DEV_KERNARG=$k timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids >> $O/gap.log; fi
done
env | grep -i "HIP\|HSA\|ROC" >> $O/gap.log
cat $O/gap.log
