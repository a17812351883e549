#!/bin/bash
# full check: gpu tests, smoke, bench, layer kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03t}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed" $O/pytest.log | tail -3; tail -2 $O/smoke.log; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['quantize_kernel']); print({k:v for k,v in d['decode'].items() if 'linears' not in k and k!='scope'}); print(d['variants']['layer_prefill']); print(d['variants']['layer_prefill_full']); print(d['variants']['layer_prefill_full_w4a8'])"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pl -o p -- python $R/tools/prof_layer.py > $O/prof_layer.log 2>&1
python $R/tools/pmc_summary.py /tmp/pl/p_results.db 2>/dev/null | grep "mq::" > $O/layer_trace.summary.txt; cat $O/layer_trace.summary.txt
