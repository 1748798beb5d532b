mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py -m gpu -q > gpurun_out/pytest_r2k.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_r2k.log; echo "pytest rc=$rc"
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'])" || echo "FAILED $*"; }
run --workload shadow; run --workload shadow; run --workload shadow --no-sort-rays; run --workload incoherent --sort-rays
