#!/bin/bash
# round-2 experiment C: what-if stamps of the free-running kernel; new gpu tests; bench launcher + calibration workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
for t in frs frs_noa frs_now frs_nord frs_nomfma frs_mfmaonly; do
  echo "== $t" >> $O/stamps.log
  timeout 120 tools/mq_probe_$t prof 11 16 20 2>&1 | grep -v "blk100" >> $O/stamps.log
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 900 python bench.py --workload calibration --calib-samples 64 > $O/calib.json 2> $O/calib.err; echo "rc=$?" >> $O/calib.err
timeout 900 python bench.py --workload calibration --calib-samples 32 --per-channel > $O/calib_pc.json 2> $O/calib_pc.err; echo "rc=$?" >> $O/calib_pc.err
timeout 120 python bench.py --gpus 2 --steps 20 > $O/gpus2.out 2>&1; echo "rc=$?" >> $O/gpus2.out
grep -v "blk0" $O/stamps.log; tail -15 $O/pytest.log; tail -3 $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/calib.err; cat $O/calib.json; tail -3 $O/calib_pc.err; cat $O/gpus2.out
