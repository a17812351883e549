#!/bin/bash
# lease r04o: tile order vs L2 fetch bytes of the N = 2048 GEMMs; the training step after the vectorised backward
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04o; mkdir -p $OUT
cd $R
python tools/groupm_probe.py > $OUT/groupm_time.log 2>&1; cat $OUT/groupm_time.log
cd /tmp && export TMPDIR=/tmp
for g in 1 2 4 8 16; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gm_$g -o p -- python $R/tools/groupm_probe.py $g > $OUT/gm_$g.log 2>&1
  python $R/tools/pmc_summary.py /tmp/gm_$g/p_results.db 2>&1 | grep -E "fr128|FETCH" | head -8 > $OUT/gm_$g.summary.txt
  echo "== group_m $g"; cat $OUT/gm_$g.summary.txt; rm -rf /tmp/gm_$g
done
cd $R
timeout 600 python tools/train_prof.py > $OUT/train_prof.log 2>&1; head -12 $OUT/train_prof.log | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q -k "fake_quant or backward or train or lwc or grad" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
