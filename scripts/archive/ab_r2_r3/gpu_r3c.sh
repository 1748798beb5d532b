# round 3, C: the hand-written packet kernel (rtk_packet_hot.S) against the C++ one (RTK_AMD_PACKET_ASM=0); parity first
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_trace.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest_r3c.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_r3c.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|rror|assert|FAILED" gpurun_out/pytest_r3c.log | head -20; exit $rc; fi
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG $*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'], d['roofline']['wave_steps_per_64_rays'], d['config']['hit_fraction'])" || exit 1; }
{
TAG=asm run --workload coherent
RTK_AMD_PACKET_ASM=0 TAG=cpp run --workload coherent
} 2>&1 | tee gpurun_out/ab_r3c.log
