# A/B of two library builds on one box: usage r06_ab_libs.sh <tagA-or-empty> <tagB> ...   (tags under mobilequant_amd/lib/<tag>/)
cd $GRAFT_REPO_ROOT
for tag in "$@" "$@"; do
  if [ "$tag" != "prod" ]; then export MQ_LIB_PATH=mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== $tag"; python tools/r06_decode_ab.py "launches=4" 2>&1 | grep "tok/s"
done
