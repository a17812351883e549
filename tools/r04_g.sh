#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R; rm -f $O/k.log
for t in "$@"; do echo "== $t" >> $O/k.log
MQ_LIB_PATH=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so HOLE_KSWEEP=1 HOLE_BRIEF=1 timeout 300 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids >> $O/k.log; done
cut -c1-200 $O/k.log
