#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -8 $O/pytest.log; tail -2 $O/smoke.log; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']); print({k:v for k,v in d['decode'].items() if 'linears' not in k and k!='scope'}); print(d['variants']['layer_prefill']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['layer']['seconds_per_layer'])"
