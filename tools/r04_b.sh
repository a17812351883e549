#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so timeout 600 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids > $O/hole.log
cat $O/hole.log
