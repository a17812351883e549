#!/bin/bash
# lease r04q: grid-barrier cost with the microarchitecture guide's XCD-hierarchical recipe; decode stamps (per-launch timeline); the
# training step after the fill trims
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04q; mkdir -p $OUT
cd $R
hipcc --offload-arch=gfx950 -O2 tools/barrier_probe.cpp -o /tmp/barrier_probe 2>/dev/null
timeout 120 /tmp/barrier_probe 2000 > $OUT/grid_barrier_probe.log 2>&1; cat $OUT/grid_barrier_probe.log
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', extra_flags=['-DMQ_DECODE_STAMPS'])" > $OUT/build_stamps.log 2>&1
MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 CONTEXT=256 timeout 300 python tools/decode_stamps.py > $OUT/decode_stamps_w8.log 2>&1; tail -9 $OUT/decode_stamps_w8.log | cut -c1-260
MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 CONTEXT=256 WBITS=4 timeout 300 python tools/decode_stamps.py > $OUT/decode_stamps_w4.log 2>&1; tail -9 $OUT/decode_stamps_w4.log | cut -c1-160
timeout 600 python tools/train_prof.py > $OUT/train_prof.log 2>&1; grep -E "bench_train_step|device kernels" $OUT/train_prof.log | cut -c1-200
