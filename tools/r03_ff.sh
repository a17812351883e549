#!/bin/bash
# A/B on one box: q rows prepared by the prep kernel (0) or inside the attention workgroups (1); index input as in the fused layer
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for on in 0 1; do
  MQ_ATT_IDX=1 MQ_ATT_FUSED_Q=$on MQ_ATT_ITERS=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa$on -o p -- python $GRAFT_REPO_ROOT/tools/prof_attention.py 2>&1 | grep "attention op"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pa$on/p_results.db 2>/dev/null | grep "attention_"; rm -rf /tmp/pa$on
done; done
