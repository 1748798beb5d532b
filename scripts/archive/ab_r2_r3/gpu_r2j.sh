mkdir -p gpurun_out
RTK_AMD_LIB=$PWD/build/libs/librtk_t64.so timeout -k 10 600 python -m pytest tests/test_gpu_trace.py tests/test_gpu_build.py -m gpu -q -k "not slow" > gpurun_out/pytest_r2j.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_r2j.log; echo "pytest(t64) rc=$rc"
run() { lib=$1; shift; RTK_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('lib=%s %s' % ('$lib'.split('/')[-1] or 'default', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], 'build', d['config']['bvh_build_ms_device_resident_mesh'])" || echo "FAILED $lib $*"; }
for wl in coherent incoherent shadow; do for lib in "" $PWD/build/libs/librtk_t64.so; do run "$lib" --workload $wl; done; done 2>&1 | tee gpurun_out/ab_r2j.log
run "" --workload shadow --sort-rays | tee -a gpurun_out/ab_r2j.log
run "$PWD/build/libs/librtk_t64.so" --workload shadow --sort-rays | tee -a gpurun_out/ab_r2j.log
