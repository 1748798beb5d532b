mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/pytest_r2h.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_r2h.log; echo "pytest rc=$rc"
run() { tb=$1; shift; RTK_AMD_TILE_BLOCKS=$tb timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('tile_blocks=%s %s' % ('$tb', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['wave_steps_per_64_rays'])" || echo "FAILED $tb $*"; }
for rep in 1 2; do for tb in 0 1; do run $tb --workload coherent; done; done 2>&1 | tee gpurun_out/ab_r2h.log
run 0 --workload coherent --no-packet | tee -a gpurun_out/ab_r2h.log
run 1 --workload coherent --no-packet | tee -a gpurun_out/ab_r2h.log
