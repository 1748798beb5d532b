timeout -k 10 600 python -m pytest tests/test_gpu_trace.py -m gpu -q -x -k "launch_modes" 2>&1 | tail -3 || exit 1
run() { timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], 'Mrays/s', d['config']['hit_fraction'])" || exit 1; }
for ne in 1 16 32 64; do for rf in 8 16 64; do run --workload incoherent --node-exit $ne --refill-min $rf; done; done
for ne in 1 16 64; do for rf in 64 32; do run --workload coherent --node-exit $ne --refill-min $rf; done; done
for ne in 1 64; do for rf in 64 16; do run --workload coherent --no-tiling --node-exit $ne --refill-min $rf; done; done
