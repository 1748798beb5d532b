mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_build.py -m gpu -q -x > gpurun_out/pytest_build.log 2>&1; rc=$?; tail -40 gpurun_out/pytest_build.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; rc=$?; tail -5 gpurun_out/bench2.err; cat gpurun_out/bench2.json; echo "bench rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
