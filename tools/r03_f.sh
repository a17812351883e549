#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03r}; mkdir -p $O; cd $R
for cfg in "0 0" "1 0" "1 1.0" "1 1.5" "1 2.0" "0.7 1.0" "0.7 1.5" "0.5 1.5"; do set -- $cfg
echo "== PREFETCH=$1 DELAY=$2 $(PREFETCH=$1 PFDELAY=$2 LAYERS=22 timeout 600 python tools/prof_decode_engine.py 2>&1 | grep 'graph ms/token')"
done
export MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so
echo "== PREFETCH=1 1.5"; PREFETCH=1 PFDELAY=1.5 LAYERS=6 timeout 600 python tools/decode_stamps.py 2>&1 | grep "gemv\|attention\|graph" | cut -c1-110
