cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "decode or generate or engine or full_depth or perplexity or recipes" 2>&1 | tail -4
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, bench
dev = torch.device("cuda:0")
with torch.no_grad():
    for wb in (8, 4):
        r = bench.bench_decode_full(dev, wbits=wb)
        print("decode W%dA8" % wb, r["decode_tok_s"], "tok/s", r["ms_per_token"], "ms")
PY
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', extra_flags=['-DMQ_DECODE_STAMPS'])" > /dev/null 2>&1
MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 CONTEXT=256 WBITS=8 timeout 300 python tools/decode_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmean" | cut -c1-200
