mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2b.log 2>&1; rc=$?; tail -25 gpurun_out/pytest_r2b.log; echo "pytest rc=$rc"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; echo "bench rc=$?"; cat gpurun_out/bench_r2b.json | cut -c1-2500
