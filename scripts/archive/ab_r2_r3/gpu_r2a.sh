# round 2, first GPU check: new builder + validator + filters + thread safety
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r2a.log 2>&1; rc=$?; tail -15 gpurun_out/pytest_r2a.log; echo "pytest rc=$rc"
timeout -k 10 300 python scripts/build_timing.py > gpurun_out/build_timing_r2a.log 2>&1; echo "timing rc=$?"; cat gpurun_out/build_timing_r2a.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; echo "bench rc=$?"; cat gpurun_out/bench_r2a.json | cut -c1-1500
