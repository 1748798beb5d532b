# round 5: build tests, then timings at 1M and 10M
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sizes.py -m gpu -q -x > gpurun_out/r5/pytest_f.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_f.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py 1000000 10000000 2>&1 | grep -E "device-resident|collapse  |refit  |finish  |emit  |sort  "
