for cn in 0.25 0.5 0.75 1 1.5; do for ml in 4 8; do
  RTK_AMD_SAH_CN=$cn RTK_AMD_MAX_LEAF=$ml timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload coherent 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('packet cn=$cn ml=$ml', d['value'], 'Mrays/s', d['roofline']['visits_per_ray'], d['roofline']['wave_steps_per_64_rays'])" || exit 1
done; done
