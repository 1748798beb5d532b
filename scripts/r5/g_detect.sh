mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_api_rows.py -m gpu -q -x > gpurun_out/r5/pytest_g.log 2>&1; rc=$?; tail -4 gpurun_out/r5/pytest_g.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/r5/pytest_g.log | head -20; exit $rc; fi
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], d['config']['without_image_hint'])"
