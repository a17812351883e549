for i in 1 2 3 4; do
echo NEW; timeout 120 python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['quantize_kernel']['avg_launch_us'])"
echo OLD; MQ_LIB_PATH=tools/ab_old/libmobilequant_amd.so timeout 120 python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['quantize_kernel']['avg_launch_us'])"
done
