#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03j}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "decode" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|error|assert" $O/pytest.log | tail -12
for sp in 1 4; do
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so SPLITS=$sp LAYERS=6 timeout 600 python tools/decode_stamps.py > $O/stamps_w8_s$sp.log 2>&1; tail -8 $O/stamps_w8_s$sp.log
SPLITS=$sp LAYERS=22 timeout 600 python tools/prof_decode_engine.py 2>&1 | grep "ms/token"
done
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so SPLITS=4 CONTEXT=1000 LAYERS=6 timeout 600 python tools/decode_stamps.py 2>&1 | grep "attention\|graph"
MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so SPLITS=1 CONTEXT=1000 LAYERS=6 timeout 600 python tools/decode_stamps.py 2>&1 | grep "attention\|graph"
