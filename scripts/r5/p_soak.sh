# round 5: soak of the device build after the stream changes (tiles beside pass 2 and the top collapse): sizes / kinds twice with
# equal hashes, then degenerate mixes with the tile collapse forced on small scenes
mkdir -p gpurun_out/r5
timeout -k 10 500 python scripts/soak_refit.py 1 > gpurun_out/r5/soak_refit.log 2>&1; rc=$?; tail -3 gpurun_out/r5/soak_refit.log; echo "soak rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
RTK_AMD_TILE_COLLAPSE_MIN=0 timeout -k 10 500 python scripts/fuzz_builds.py 300 120 > gpurun_out/r5/fuzz_tile.log 2>&1; rc=$?; tail -3 gpurun_out/r5/fuzz_tile.log; echo "fuzz rc=$rc"
exit $rc
