# round 5: A/B of build variants (variants/libs/librtk_<name>.so; "base" = the tree's library), 10M triangles, twice each
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset RTK_AMD_LIB; else export RTK_AMD_LIB=$PWD/variants/libs/librtk_$v.so; fi
  echo "== $v: $(timeout -k 10 300 python scripts/build_timing.py 10000000 2>&1 | grep -E 'device-resident' | tail -1)" || exit 1
done
done
