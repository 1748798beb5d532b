# A/B of the ray-reordering key: origin cell in the batch's origin bounds (RTK_AMD_SORT_KEY=0) vs the cell of the
# entry point into the scene's bounds (1), at several cell resolutions, with and without the direction octant.
run() { timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('key=$SK bits=$SB oct=$SO $*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" || exit 1; }
mkdir -p gpurun_out
{
SK=- SB=- SO=- run --workload incoherent
SK=- SB=- SO=- run --workload shadow --no-sort-rays
for SK in 0 1; do for SB in 4 5 6 7; do for SO in 0 1; do export RTK_AMD_SORT_KEY=$SK RTK_AMD_SORT_CELL_BITS=$SB RTK_AMD_SORT_OCTANT=$SO SK SB SO
run --workload incoherent --sort-rays
run --workload shadow --sort-rays
done; done; done
} 2>&1 | tee gpurun_out/ab_r2q.log
