# node_exit / refill_min re-sweep after the one-entry-per-trip pop
run() { timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s', r['kernel_ms'], r['wave_steps_per_64_rays'])" || exit 1; }
mkdir -p gpurun_out
{
for ne in 20 24 28 32 36 40 48; do run --workload incoherent --node-exit $ne; done
for rm in 4 6 12 16; do run --workload incoherent --refill-min $rm; done
for ne in 24 28 36 40; do run --workload shadow --node-exit $ne; done
for rm in 4 12; do run --workload shadow --refill-min $rm; done
} 2>&1 | tee gpurun_out/ab_r2t.log
