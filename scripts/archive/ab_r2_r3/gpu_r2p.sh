run() { timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['wave_steps_per_64_rays'])" || echo "FAILED $*"; }
for ne in 28 32 36 40; do for rm in 6 8 12; do run --workload incoherent --refill-min $rm --node-exit $ne; done; done 2>&1 | tee gpurun_out/ab_r2p.log
for ne in 28 32 36 40; do run --workload shadow --node-exit $ne; done 2>&1 | tee -a gpurun_out/ab_r2p.log
for ne in 24 32 40; do run --workload coherent --no-packet --node-exit $ne; done 2>&1 | tee -a gpurun_out/ab_r2p.log
