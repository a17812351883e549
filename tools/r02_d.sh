#!/bin/bash
# round-2 experiment D: where does the wall time of the free-running kernel go?  realtime stamps, DVFS what-if (operand fill)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
timeout 120 tools/mq_probe_frs prof 11 16 20 2>&1 | grep -v "blk100" >> $O/stamps.log
for f in zero small; do
  echo "== fill $f" >> $O/stamps.log
  MQ_PROBE_FILL=$f timeout 120 tools/mq_probe_frs prof 11 16 20 2>&1 | grep -v "blk" >> $O/stamps.log
  MQ_PROBE_FILL=$f timeout 120 tools/mq_probe prof 11 0 50 2>&1 | grep prof >> $O/stamps.log
  MQ_PROBE_FILL=$f timeout 120 tools/mq_probe prof 9 0 50 2>&1 | grep prof >> $O/stamps.log
done
echo "== K sweep (fixed cost)" >> $O/stamps.log
for K in 768 1024 2048 4096; do timeout 120 tools/mq_probe prof 11 0 50 2048 5632 $K 2>&1 | grep "prof t" >> $O/stamps.log; done
cat $O/stamps.log
