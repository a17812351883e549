#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
for t in frs st_wb st_sc1 st_sc0sc1 st_none; do
  echo "== $t" >> $O/store_policy.log
  MQ_LIB_PATH=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so DVFS_FILLS=gauss DVFS_VARIANTS=11 DVFS_EAGER=1 timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids >> $O/store_policy.log
done
MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so DVFS_FILLS=gauss DVFS_VARIANTS=11 DVFS_POWER=1 timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids >> $O/power.log
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/store_policy.log $O/power.log; tail -15 $O/pytest.log
