#!/bin/bash
# round 3: the sticky form of the ray pool: tests (both forms), then the two per-lane workloads with and without re-ordering
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for st in 1 0; do
  RTK_AMD_POOL_STICKY=$st timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r3o_pytest_$st.log 2>&1; rc=$?; echo "sticky=$st: $(tail -1 gpurun_out/r3o_pytest_$st.log)"; [ $rc -eq 0 ] || { tail -20 gpurun_out/r3o_pytest_$st.log; exit 1; }
done
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*: value %.1f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
export RTK_AMD_POOL=1
run --workload incoherent
run --workload incoherent --sort-rays
run --workload shadow
RTK_AMD_POOL=0 run --workload incoherent
RTK_AMD_POOL=0 run --workload shadow
