#!/bin/bash
# round 3, step l: BASELINE.json configs[4] on one GPU, 512 samples: stub-GEMM mode (hot path only) and the full fp32 model
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python bench.py --workload calibration --calib-stub-gemm > $O/calibration_512_stub_gemm.json 2> $O/stub.err; echo "rc=$?" >> $O/stub.err
timeout 900 python bench.py --workload calibration --calib-stub-gemm --per-channel --no-cpu-baseline > $O/calibration_512_stub_gemm_per_channel.json 2>> $O/stub.err; echo "rc=$?" >> $O/stub.err
timeout 1200 python bench.py --workload calibration --no-cpu-baseline > $O/calibration_512_fp32_model.json 2> $O/full.err; echo "rc=$?" >> $O/full.err
tail -3 $O/stub.err $O/full.err
for f in $O/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], d['breakdown'], d.get('cpu_baseline'))"; done
