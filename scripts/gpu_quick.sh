mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py -m gpu -q -x > gpurun_out/pytest_quick.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_quick.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|rror" gpurun_out/pytest_quick.log | head; exit $rc; fi
run() { timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'], d['config']['hit_fraction'])" || exit 1; }
run --workload coherent
run --workload incoherent
run --workload incoherent --sort-rays
run --workload shadow
run --workload shadow --sort-rays
