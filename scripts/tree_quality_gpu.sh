# Tree quality on the GPU itself: the same kernels on the device LBVH and on a full top-down binned-SAH
# tree with one-triangle leaves (the product's CPU task-graph builder, blob uploaded). Prints value, node / leaf visits per ray
# and wave-level steps per 64 rays.  bash scripts/tree_quality_gpu.sh > profiles/r02_tree_quality_gpu.log
export RTK_AMD_CPU_SAH_SPLIT_COST=0.5 RTK_AMD_CPU_LEAF_MIN=1
for wl in coherent incoherent shadow; do
  for bvh in device cpu-sah oracle-blob; do      # oracle-blob: the CPU oracle's SAH tree as it is (leaves of ~3 triangles)
    timeout -k 10 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl --bvh $bvh 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl', '$bvh', d['value'], 'Mrays/s', 'nodes', d['config']['bvh_nodes'], 'visits', r['visits_per_ray'], 'wave steps', r['wave_steps_per_64_rays'])"
  done
done
