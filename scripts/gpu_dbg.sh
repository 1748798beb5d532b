mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_build.py -m gpu -q -x -s -k "edge_scene or tiny" > gpurun_out/pytest_dbg.log 2>&1; rc=$?; head -30 gpurun_out/pytest_dbg.log; echo "pytest rc=$rc"
