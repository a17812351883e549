#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03c}; mkdir -p $O; cd $R
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 timeout 600 python tools/decode_stamps.py > $O/stamps_w8.log 2>&1; cat $O/stamps_w8.log | tail -9
