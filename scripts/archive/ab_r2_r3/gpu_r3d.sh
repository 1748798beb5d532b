# round 3, D: tile-local collapse in the device build; build tests, then timings and the three bench lines
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sizes.py tests/test_gpu_trace.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest_r3d.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_r3d.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|rror|assert|FAILED" gpurun_out/pytest_r3d.log | head -20; exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py 2>&1 | grep -v amdgpu.ids | tail -30
