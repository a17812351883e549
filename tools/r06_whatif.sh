cd $GRAFT_REPO_ROOT
python - <<'PY' > /dev/null 2>&1
from mobilequant_amd import build
build.build(force=True, tag='nokeys', only=['mq_decode.hip'], extra_flags=['-DMQ_AO_WHATIF_NOKEYS'])
build.build(force=True, tag='nokv', only=['mq_decode.hip'], extra_flags=['-DMQ_AO_WHATIF_NOKEYS', '-DMQ_AO_WHATIF_NOVALUES'])
PY
for tag in "" nokeys nokv ""; do
  if [ -n "$tag" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== ${tag:-production}"
  python tools/r06_decode_ab.py "launches=4" 2>&1 | grep "tok/s"
done
