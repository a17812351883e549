#!/bin/bash
# round-2 experiment B: the free-running generated kernel (variant 11): host-checked correctness, timing, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
for i in 1 2; do
  timeout 120 tools/mq_probe prof 11 0 50 >> $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
  timeout 120 tools/mq_probe prof 9 0 50 >> $O/probe.log 2>&1
done
timeout 120 tools/mq_probe_frstamp prof 11 16 20 >> $O/stamps.log 2>&1; echo "rc=$?" >> $O/stamps.log
timeout 120 tools/mq_probe prof 11 0 50 2048 2112 5632 >> $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
timeout 120 tools/mq_probe prof 11 0 50 2000 5632 2048 >> $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
timeout 120 tools/mq_probe prof 11 0 50 2048 5632 768 4 >> $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
cat $O/probe.log; grep -v blk $O/stamps.log; grep "blk0 " $O/stamps.log
