#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -20 $O/pytest.log; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']); print(d['variants']['layer_prefill'])"
