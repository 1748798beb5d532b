# round 5: rtk_packet_beam2 after the packed triangle arithmetic and the one-divide set-up: parity tests of the packet paths, then the frame rate
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_lane_asm.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r5/pytest_b.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_b.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], 'frac', r['frac'], r.get('timed_kernel_steps'))"; done
