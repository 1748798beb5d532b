#!/bin/bash
# round 4: axis-uniform leaf loops; 15 LDS entries / 5 workgroups per CU against 12 entries / 6 workgroups
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_lane_asm.py -m gpu -q -x > gpurun_out/r4e_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r4e_pytest.log; [ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%s' % os.path.basename(os.environ.get('RTK_AMD_LIB','default')), '$*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" || exit 1; }
run --workload incoherent --sort-rays
run --workload incoherent
run --workload shadow
export RTK_AMD_LIB=$PWD/variants/libs/librtk_l12.so RTK_AMD_LANE_LDS=12 RTK_AMD_LANE_BLOCKS=6
run --workload incoherent --sort-rays
run --workload incoherent
run --workload shadow
