#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03o}; mkdir -p $O; cd $R
export MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so
for L in 2 22; do for W in 8 4; do
echo "== LAYERS=$L NOHEAD W$W"; NOHEAD=1 LAYERS=$L WBITS=$W timeout 600 python tools/decode_stamps.py 2>&1 | grep "gemv\|attention\|graph" | cut -c1-110
done; done
