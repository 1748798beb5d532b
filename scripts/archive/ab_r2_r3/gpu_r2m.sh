mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2m.log 2>&1; rc=$?; tail -8 gpurun_out/pytest_r2m.log; echo "pytest rc=$rc"
