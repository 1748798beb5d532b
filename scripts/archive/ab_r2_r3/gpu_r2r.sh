# SAH leaf rule of the device build (node cost cn, max leaf size) against the per-lane workloads: a node step costs
# the per-lane kernel ~200 VALU instructions at 67 % lane use, a triangle step ~130 at 37 %.
run() { timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('cn=$RTK_AMD_SAH_CN ml=$RTK_AMD_MAX_LEAF $*', d['value'], 'Mrays/s', r['kernel_ms'], r['visits_per_ray'], r['wave_steps_per_64_rays'])" || exit 1; }
mkdir -p gpurun_out
{
for cn in 0.5 1 1.5 2 3; do for ml in 4 8; do export RTK_AMD_SAH_CN=$cn RTK_AMD_MAX_LEAF=$ml
run --workload incoherent
run --workload shadow
run --workload coherent
done; done
} 2>&1 | tee gpurun_out/ab_r2r.log
