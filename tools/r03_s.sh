#!/bin/bash
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pl -o p -- python $GRAFT_REPO_ROOT/tools/prof_layer.py 2>&1 | tail -1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pl/p_results.db 2>/dev/null | grep "gated_lookup\|fr128\|rmsnorm\|frg\|fr_kernel\|norm_tiled\|attention"
