#!/bin/bash
# round-2 experiment A: sanity of the epilogue refactor + stamps of the generated-ISA loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for i in 1 2; do
  tools/mq_probe prof 9 0 50 >> $O/probe.log 2>&1
  tools/mq_probe prof 7 0 50 >> $O/probe.log 2>&1
done
tools/mq_probe_ablate prof 9 16 20 >> $O/stamps.log 2>&1
tools/mq_probe_ablate prof 7 16 20 >> $O/stamps.log 2>&1
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; cat $O/probe.log $O/stamps.log; cut -c1-600 $O/bench.json
