mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -25 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --bvh oracle-blob --steps 5 --warmup 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; rc=$?; tail -5 gpurun_out/bench1.err; cat gpurun_out/bench1.json; echo "bench rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --bvh oracle-blob --steps 5 --warmup 1 --workload incoherent --no-cpu-baseline > gpurun_out/bench1_inc.json 2> gpurun_out/bench1_inc.err; rc=$?; cat gpurun_out/bench1_inc.json; echo "bench-inc rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
