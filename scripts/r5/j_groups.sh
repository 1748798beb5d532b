# round 5: full groups of four on the hand-written kernels: the GPU tests that trace blobs with the reference's leaf sizes, then
# config 2 / 3 on the CPU task builder's SAH tree against the device build
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests/test_gpu_trace.py tests/test_gpu_lane_asm.py tests/test_gpu_api_rows.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r5/pytest_j.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_j.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/r5/pytest_j.log | head -20; exit $rc; fi
for wl in coherent incoherent; do for bvh in device cpu-sah; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --workload $wl --bvh $bvh 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $bvh', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], d['roofline'].get('visits_per_ray'))"; done; done
