# A/B: set a leaf aside and keep descending (RTK_POSTPONE=1 build in build/libs) vs the default library
run() { timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$TAG $*', d['value'], 'Mrays/s', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'], 'parity', d.get('parity'))" || exit 1; }
mkdir -p gpurun_out
{
TAG=default run --workload incoherent
TAG=default run --workload shadow
export RTK_AMD_LIB=$PWD/build/libs/librtk_postpone.so
timeout -k 10 300 python -m pytest tests/test_gpu_trace.py -m gpu -x -q 2>&1 | tail -2
for ne in 24 32 40 48; do TAG=postpone run --workload incoherent --node-exit $ne; done
for ne in 24 32 40 48; do TAG=postpone run --workload shadow --node-exit $ne; done
TAG=postpone run --workload coherent --no-packet
} 2>&1 | tee gpurun_out/ab_r2u.log
