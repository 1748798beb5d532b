#!/bin/bash
# round 3: any-hit kernel without pop culling, re-ordering at 2^7 cells: tests, then a sweep of the wave-scheduling options on the shadow batch
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_api_rows.py -x -q -m gpu > gpurun_out/r3l_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3l_pytest.log; [ $rc -eq 0 ] || exit 1
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*: value %.1f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
run --workload shadow
for ne in 24 40 48; do run --workload shadow --node-exit $ne; done
for rm in 4 16 24; do run --workload shadow --refill-min $rm; done
run --workload incoherent
for ne in 24 40; do run --workload incoherent --node-exit $ne; done
