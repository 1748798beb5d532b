#!/bin/bash
# round 4: first run of the hand-written per-lane kernels: their tests, then A/B against the C++ kernel
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_lane_asm.py -m gpu -q -x > gpurun_out/r4a_pytest.log 2>&1; rc=$?; tail -15 gpurun_out/r4a_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('asm=%s' % os.environ.get('RTK_AMD_LANE_ASM','1'), '$*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'], d['config']['hit_fraction'])" || exit 1; }
for asm in 1 0; do
  export RTK_AMD_LANE_ASM=$asm
  run --workload incoherent
  run --workload incoherent --sort-rays
  run --workload shadow
  run --workload shadow --no-sort-rays
done
