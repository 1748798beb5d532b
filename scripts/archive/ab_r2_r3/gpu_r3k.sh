#!/bin/bash
# round 3: per-lane kernel iteration: trace tests, then the three workloads
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pool.py tests/test_gpu_api_rows.py -x -q -m gpu > gpurun_out/r3k_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3k_pytest.log; [ $rc -eq 0 ] || exit 1
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*: value %.1f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
run --workload incoherent
RTK_AMD_SORT_CELL_BITS=6 run --workload incoherent --sort-rays
RTK_AMD_SORT_CELL_BITS=5 run --workload shadow
RTK_AMD_SORT_CELL_BITS=7 run --workload shadow
run --workload coherent
