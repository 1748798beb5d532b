# round 5: the largest level the one-workgroup collapse kernel takes (64: round 2's choice), 1M and 10M triangles
for rep in 1 2; do for sj in 64 256 1024; do
  echo "small_jobs=$sj $(RTK_AMD_SMALL_JOBS=$sj timeout -k 10 300 python scripts/build_timing.py 1000000 10000000 2>&1 | grep -E 'device-resident' | tr '\n' ' ')" || exit 1
done; done
