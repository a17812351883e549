#!/bin/bash
# A/B on one box: default library against an experiment build in lib/$1 (MQ_LIB_PATH), attention at the layer's shape
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for lib in "" $1; do
  if [ -n "$lib" ]; then export MQ_LIB_PATH=$GRAFT_REPO_ROOT/mobilequant_amd/lib/$lib/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "lib=${lib:-default}"
  MQ_ATT_IDX=1 MQ_ATT_ITERS=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o p -- python $GRAFT_REPO_ROOT/tools/prof_attention.py 2>&1 | grep "attention op"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pa/p_results.db 2>/dev/null | grep "attention_quant"; rm -rf /tmp/pa
done; done
