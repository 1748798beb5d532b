# packet kernel A/B: default run of the headline, twice, after the parity tests
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s', r['kernel_ms'])" || exit 1; }
mkdir -p gpurun_out
{
run --workload coherent
run --workload coherent
run --workload coherent
} 2>&1 | tee gpurun_out/ab_r2v.log
