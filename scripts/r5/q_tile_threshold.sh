# round 5: where tile mode starts to pay after the stream changes: level-by-level (huge threshold) vs tile mode (0) per size
for n in 500000 1000000 2000000 3000000; do
  for m in 1099511627776 0; do
    echo "n=$n tile_min=$m: $(RTK_AMD_TILE_COLLAPSE_MIN=$m timeout -k 10 300 python scripts/build_timing.py $n 2>&1 | grep -E 'device-resident' | tail -1)" || exit 1
  done
done
