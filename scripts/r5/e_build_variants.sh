# round 5: stage timings of the 10M-triangle build with parts of k_collapse_tile compiled out (variant libraries; the scenes they
# make are NOT valid: timing experiments only)
for v in "" "$@"; do
  if [ -n "$v" ]; then export RTK_AMD_LIB=$PWD/variants/libs/librtk_$v.so; fi
  echo "== ${v:-base}"
  timeout -k 10 200 python scripts/build_timing.py 10000000 2>&1 | grep -E "device-resident|collapse  |refit  |emit  "
done
