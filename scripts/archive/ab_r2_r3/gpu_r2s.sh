# quick three-workload check of the default library (plus unsorted shadow)
run() { timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'])" || exit 1; }
mkdir -p gpurun_out
{
run --workload incoherent
run --workload shadow --no-sort-rays
run --workload shadow
run --workload coherent
run --workload coherent --no-packet
} 2>&1 | tee gpurun_out/ab_r2s.log
