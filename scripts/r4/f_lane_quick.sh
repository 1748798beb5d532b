#!/bin/bash
# round 4: lane-kernel tests and the three per-lane workload rates
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_lane_asm.py -m gpu -q -x > gpurun_out/r4f_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r4f_pytest.log; [ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" || exit 1; }
run --workload incoherent --sort-rays
run --workload incoherent
run --workload shadow
run --workload shadow --no-sort-rays
