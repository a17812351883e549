#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so timeout 300 python tools/dvfs_probe.py > $O/dvfs_stamped.log 2>&1
timeout 300 python tools/dvfs_probe.py > $O/dvfs_prod.log 2>&1
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > $O/smi.log
cat $O/dvfs_stamped.log $O/dvfs_prod.log $O/smi.log
