mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$*', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], r['wave_steps_per_64_rays'])" || echo "FAILED $*"; }
for wl in incoherent; do
  for rm in 4 8 16 32; do for ne in 8 16 24 32 48; do run --workload $wl --refill-min $rm --node-exit $ne; done; done
done 2>&1 | tee gpurun_out/ab_r2o.log
for ne in 8 16 24 32 48; do run --workload shadow --node-exit $ne; done 2>&1 | tee -a gpurun_out/ab_r2o.log
for bpc in 2 3 4 5; do run --workload incoherent --blocks-per-cu $bpc; done 2>&1 | tee -a gpurun_out/ab_r2o.log
