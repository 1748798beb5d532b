# round 5: where a tile's time goes in k_collapse_tile (instrumented builds, variants/libs/librtk_<name>.so, -DRTK_TILE_PHASES ...)
for v in "$@"; do
  echo "== $v"
  RTK_AMD_LIB=$PWD/variants/libs/librtk_$v.so timeout -k 10 300 python scripts/build_timing.py 10000000 2>&1 | grep -E "tile phases|device-resident|collapse  " | tail -4 || exit 1
done
