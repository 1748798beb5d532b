mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/pytest_r2f.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_r2f.log; echo "pytest rc=$rc"
run() { qn=$1; shift; RTK_AMD_QNODES=$qn timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('qnodes=%s %s' % ('$qn', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'])" || echo "FAILED $qn $*"; }
for wl in incoherent shadow; do for qn in 0 1; do run $qn --workload $wl; done; done 2>&1 | tee gpurun_out/ab_r2f.log
run 1 --workload coherent --no-packet | tee -a gpurun_out/ab_r2f.log
run 1 --workload incoherent --blocks-per-cu 4 | tee -a gpurun_out/ab_r2f.log
