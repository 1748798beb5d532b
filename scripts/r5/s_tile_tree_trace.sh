# round 5: does the tile-collapsed tree (2 % more nodes) cost trace time on the 1M-triangle scene? coherent and incoherent, both trees
for m in 1099511627776 0; do
  for wl in coherent incoherent; do
    RTK_AMD_TILE_COLLAPSE_MIN=$m timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-workloads --workload $wl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile_min=$m $wl', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'], 'build', d['build']['ms'], 'nodes', d.get('scene',{}).get('num_nodes'))" || exit 1
  done
done
