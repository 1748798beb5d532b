# round 5: where the host is during a build (no synchronisation): is the enqueueing thread ahead of the GPU?
for n in "$@"; do
  RTK_AMD_BUILD_HOSTTIME=1 timeout -k 10 300 python scripts/build_timing.py $n 2>&1 | grep -E "build host|device-resident" | head -48 | tail -20
done
