# round 3, A: packet kernel with the one-fma slab test (margins folded into the constants) and the nodes' static child order,
# against the round-2 kernel (build/libs/librtk_base.so); parity first
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest_r3a.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_r3a.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|rror|assert" gpurun_out/pytest_r3a.log | head -20; exit $rc; fi
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG $*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'], d['roofline']['wave_steps_per_64_rays'], d['config']['hit_fraction'])" || exit 1; }
{
TAG=new run --workload coherent
export RTK_AMD_LIB=$PWD/build/libs/librtk_w6.so; TAG=new_6waves run --workload coherent
export RTK_AMD_LIB=$PWD/build/libs/librtk_base.so; TAG=base run --workload coherent
unset RTK_AMD_LIB
TAG=new run --workload incoherent
TAG=new run --workload shadow
} 2>&1 | tee gpurun_out/ab_r3a.log
