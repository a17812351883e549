#!/bin/bash
# decode engine kernel trace, 22 layers, int8 and packed 4-bit weights on one box
cd /tmp; export TMPDIR=/tmp
for wb in 8 4; do
  echo "WBITS=$wb"
  WBITS=$wb LAYERS=22 PREFETCH=0.5 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python $GRAFT_REPO_ROOT/tools/prof_decode_engine.py 2>&1 | grep "ms/token"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pd/p_results.db 2>/dev/null | grep "mq::decode" | tee $GRAFT_REPO_ROOT/gpurun_out/decode_engine_trace_w$wb.summary.txt; rm -rf /tmp/pd
done
