# round 5: the SAH constants of the device build against the hand-written kernels (they were swept in round 1 against the C++ kernels)
for ml in 3 4 8; do
for cn in 0.25 0.5 1.0 2.0 4.0; do
  for wl in coherent incoherent; do
    RTK_AMD_SAH_CN=$cn RTK_AMD_MAX_LEAF=$ml timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-workloads --workload $wl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('max_leaf=$ml cn=$cn $wl', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], 'build', d['build']['ms'])" || exit 1
  done
done
done
