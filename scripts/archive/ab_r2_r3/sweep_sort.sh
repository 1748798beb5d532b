run() { timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$SB $SO $*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" || exit 1; }
for SB in 4 5 6 7; do for SO in 0 1; do export RTK_AMD_SORT_CELL_BITS=$SB RTK_AMD_SORT_OCTANT=$SO SB SO
run --workload shadow --sort-rays
run --workload incoherent --sort-rays
done; done
