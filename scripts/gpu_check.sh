# full GPU test suite + the two bench workloads; used after every kernel change
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -6 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|Error|error" gpurun_out/pytest_gpu.log | head; exit $rc; fi
for wl in coherent incoherent; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $wl $BENCH_EXTRA 2>gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], d['roofline']['visits_per_ray'], 'frac', d['roofline']['frac'])" || { tail -5 gpurun_out/bench_$wl.err; exit 1; }
done
