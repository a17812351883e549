#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so
for W in 8 4; do
echo "== LAYERS=6 W$W"; LAYERS=6 PREFETCH=0.5 WBITS=$W timeout 600 python tools/decode_stamps.py 2>&1 | grep "gemv\|attention\|graph\|head" | cut -c1-128
done
