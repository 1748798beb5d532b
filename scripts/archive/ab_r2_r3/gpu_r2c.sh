mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2c.log 2>&1; rc=$?; tail -12 gpurun_out/pytest_r2c.log; echo "pytest rc=$rc"
run() { lib=$1; pp=$2; shift 2; RTK_AMD_LIB=$lib RTK_AMD_POSTPONE=$pp timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('lib=%s postpone=%s %s' % ('$lib'.split('/')[-1] or 'default', '$pp', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'])" || echo "FAILED $lib $pp $*"; }
for wl in incoherent shadow; do
  for lib in "" $PWD/build/libs/librtk_w5.so; do
    for pp in 0 1; do run "$lib" $pp --workload $wl; done
  done
done 2>&1 | tee gpurun_out/ab_r2c.log
run "" 0 --workload coherent --no-packet | tee -a gpurun_out/ab_r2c.log
run "$PWD/build/libs/librtk_w5.so" 1 --workload coherent --no-packet | tee -a gpurun_out/ab_r2c.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; echo "bench rc=$?"; cut -c1-3000 gpurun_out/bench_r2c.json
