#!/bin/bash
# attention at the layer's shape, index input: kernel trace (two repeats)
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  MQ_ATT_IDX=1 MQ_ATT_ITERS=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o p -- python $GRAFT_REPO_ROOT/tools/prof_attention.py 2>&1 | grep "attention op"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pa/p_results.db 2>/dev/null | grep "attention_"; rm -rf /tmp/pa
done
