#!/bin/bash
# lease r04p: the fused LWC pass (f3) -- parity vs the module chain and the goldens, the training step with it, group_m FETCH passes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04p; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "lwc or train or fake_quant or grad or backward or smooth or let" > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 600 python tools/train_prof.py > $OUT/train_prof.log 2>&1; grep -E "bench_train_step|device kernels|lwc_|fake_quant" $OUT/train_prof.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for g in 1 2 4 8 16; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gm_$g -o p -- python $R/tools/groupm_probe.py $g > $OUT/gm_$g.log 2>&1
  python $R/tools/pmc_summary.py /tmp/gm_$g/p_results.db 2>&1 | grep -A1 "fr128" > $OUT/gm_$g.summary.txt
  echo "== group_m $g"; cat $OUT/gm_$g.summary.txt; rm -rf /tmp/gm_$g
done
