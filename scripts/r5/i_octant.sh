# round 5: config 3 / config 5 with the direction octant in the re-ordering key (RTK_AMD_SORT_OCTANT) and other key widths
run() { env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --workload $WL 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL $*', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'])"; }
for WL in incoherent shadow; do
  run A=1
  run RTK_AMD_SORT_OCTANT=1
  run RTK_AMD_SORT_CELL_BITS=6 RTK_AMD_SORT_OCTANT=1
  run RTK_AMD_SORT_CELL_BITS=8
done
