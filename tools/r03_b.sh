#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03b}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "decode" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|error" $O/pytest.log | tail -5
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 timeout 600 python tools/decode_stamps.py > $O/stamps_w8.log 2>&1; cat $O/stamps_w8.log | tail -9
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 WBITS=4 timeout 600 python tools/decode_stamps.py > $O/stamps_w4.log 2>&1; cat $O/stamps_w4.log | tail -9
LAYERS=22 timeout 600 python tools/prof_decode_engine.py 2>&1 | grep "ms/token"
