#!/bin/bash
# round 3: pool kernel iteration: tests, statistics build, timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r3i_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r3i_pytest.log; [ $rc -eq 0 ] || exit 1
RTK_AMD_LIB=$PWD/variants/libs/librtk_stats.so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --workload incoherent 2>&1 | grep -A1 "pool stats" | tail -2
for wl in incoherent shadow; do
  for extra in "" "--sort-rays"; do
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --workload $wl $extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl $extra: value %.1f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"
  done
done
