# round 5: the triangle records made inside k_refit_tile (default) against a pass of their own (RTK_AMD_FUSED_EMIT=0): tests, then timings
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sizes.py -m gpu -q -x > gpurun_out/r5/pytest_v.log 2>&1; rc=$?; tail -3 gpurun_out/r5/pytest_v.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do for f in 1 0; do
  echo "fused=$f $(RTK_AMD_FUSED_EMIT=$f timeout -k 10 300 python scripts/build_timing.py 1000000 10000000 2>&1 | grep -E 'device-resident' | tr '\n' ' ')" || exit 1
done; done
