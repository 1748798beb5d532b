# full GPU test suite + the bench workloads; used after every kernel change
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -6 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; if [ $rc -ne 0 ]; then grep -E "VIOLATION|Error|error" gpurun_out/pytest_gpu.log | head; exit $rc; fi
for wl in coherent incoherent shadow; do
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --workload $wl $BENCH_EXTRA 2>gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], d['roofline']['visits_per_ray'], 'frac', d['roofline']['frac'], 'build_ms', d['config']['bvh_build_ms_in_library'], d['config']['bvh_build_mtris_s'])
print('   cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k!='sample'})" || { tail -5 gpurun_out/bench_$wl.err; exit 1; }
done
