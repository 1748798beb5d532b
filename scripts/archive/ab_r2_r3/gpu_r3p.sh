#!/bin/bash
# round 3: single-launch radix passes (decoupled look-back): build and trace tests, build times, the shadow batch; then the same with the three-launch passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_api_rows.py tests/test_gpu_trace.py tests/test_gpu_sizes.py -x -q -m gpu > gpurun_out/r3p_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3p_pytest.log; [ $rc -eq 0 ] || exit 1
for os in 1 0; do
  echo "== RTK_AMD_SORT_ONESWEEP=$os"
  RTK_AMD_SORT_ONESWEEP=$os timeout -k 10 300 python scripts/build_timing.py 2>&1 | grep -E "device-resident|sort  "
  RTK_AMD_SORT_ONESWEEP=$os timeout -k 10 200 python bench.py --workload shadow --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('shadow: value %.1f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"
done
timeout -k 10 300 python scripts/soak_refit.py 1 2>&1 | tail -1
