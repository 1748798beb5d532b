mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2d.log 2>&1; rc=$?; tail -12 gpurun_out/pytest_r2d.log; echo "pytest rc=$rc"
run() { lib=$1; qn=$2; shift 2; RTK_AMD_LIB=$lib RTK_AMD_QNODES=$qn timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('lib=%s qnodes=%s %s' % ('$lib'.split('/')[-1] or 'default', '$qn', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'], 'build', d['config']['bvh_build_ms_device_resident_mesh'])" || echo "FAILED $lib $qn $*"; }
for wl in incoherent shadow; do
  for lib in "" $PWD/build/libs/librtk_w5.so; do
    for qn in 0 1; do run "$lib" $qn --workload $wl; done
  done
done 2>&1 | tee gpurun_out/ab_r2d.log
run "" 1 --workload coherent --no-packet | tee -a gpurun_out/ab_r2d.log
run "" 1 --workload incoherent --sort-rays | tee -a gpurun_out/ab_r2d.log
run "" 1 --workload shadow --sort-rays | tee -a gpurun_out/ab_r2d.log
