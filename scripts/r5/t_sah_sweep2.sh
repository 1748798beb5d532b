# round 5: finer sweep of the node constant at the small end (max_leaf 3), three workloads
for cn in 0.0 0.1 0.25 0.35 0.5; do
  for wl in coherent incoherent shadow; do
    RTK_AMD_SAH_CN=$cn timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-workloads --workload $wl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cn=$cn $wl', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], 'build', d['build']['ms'])" || exit 1
  done
done
