mkdir -p gpurun_out
for wl in coherent incoherent; do
for cn in 0.5 1 2; do for ml in 2 3 4 6 8; do
  RTK_AMD_SAH_CN=$cn RTK_AMD_MAX_LEAF=$ml timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl cn=$cn ml=$ml', d['value'], 'Mrays/s nodes', d['config']['bvh_nodes'], d['roofline']['visits_per_ray'])" || exit 1
done; done; done
