#!/bin/bash
# A/B of two builds of the library on one box: ab_bench.sh <other-lib.so> [rounds]   (run through gpurun)
OTHER=$1; N=${2:-3}
one() { python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['quantize_kernel']['avg_launch_us'])"; }
for i in $(seq $N); do
  echo "tree build:"; one
  echo "$OTHER:"; MQ_LIB_PATH=$OTHER one
done
