# round 5: the device build after a change: its tests (validator, hashes, parity on built scenes, mixed meshes), then the timings
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sizes.py tests/test_gpu_api_rows.py tests/test_c_host.py -m gpu -q -x > gpurun_out/r5/pytest_d.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_d.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py 2>&1 | grep -v "^rtk_amd build:  " | tee gpurun_out/r5/build_timing_d.log
