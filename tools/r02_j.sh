#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -30 $O/pytest.log; tail -5 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']); print(json.dumps(d['decode'], indent=1)); print(d['variants']['layer_prefill'])"
