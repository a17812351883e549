#!/bin/bash
# channel camping? the same stamped kernel on dense and on padded operand images
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R; rm -f $O/pad.log
export MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so HOLE_ONLY=xcd HOLE_BRIEF=1
for rep in 1 2; do for cfg in "0 0" "128 0" "0 1024" "128 1024" "256 2048" "64 512"; do set -- $cfg
  echo "== pad W $1 A $2" >> $O/pad.log
  HOLE_PAD_W=$1 HOLE_PAD_A=$2 timeout 300 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids >> $O/pad.log
done; done
grep "^==\|^-- GEMM(gauss\|period" $O/pad.log | paste - - - - | cut -c1-330
