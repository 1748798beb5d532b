#!/bin/bash
# round 4: the whole GPU suite, then the default bench line
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4g_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r4g_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || { grep -E "Error|FAILED|assert" gpurun_out/r4g_pytest.log | head -20; exit $rc; }
timeout -k 10 400 python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r4g_bench.json
