#!/bin/bash
# round 4: shared entry points of the packet kernel: tests, then A/B and the list-size target
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_lane_asm.py tests/test_gpu_trace.py -m gpu -q -x > gpurun_out/r4h_pytest.log 2>&1; rc=$?; tail -12 gpurun_out/r4h_pytest.log; [ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 150 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('entries=%s target=%s' % (os.environ.get('RTK_AMD_PACKET_ENTRIES','1'), os.environ.get('RTK_AMD_ENTRY_TARGET','20')), '$*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'], d['roofline']['visits_per_ray'])" || exit 1; }
RTK_AMD_PACKET_ENTRIES=0 run
for t in 20 24 28 32 36; do RTK_AMD_ENTRY_TARGET=$t run; done
