cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', only=['mq_decode.hip'], extra_flags=['-DMQ_DECODE_STAMPS'])" > /dev/null 2>&1
for L in 4 5; do
MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAUNCHES=$L LAYERS=6 CONTEXT=256 WBITS=8 timeout 300 python tools/decode_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmean" | cut -c1-260 > gpurun_out/r06/decode_stamps_L$L.log
cat gpurun_out/r06/decode_stamps_L$L.log
done
