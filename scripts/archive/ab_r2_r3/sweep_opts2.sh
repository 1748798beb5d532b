run() { timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], 'Mrays/s')" || exit 1; }
for ne in 32 64; do for rf in 4 8 16 24; do run --workload incoherent --node-exit $ne --refill-min $rf; done; done
for ne in 1 64; do for rf in 64 16; do run --workload coherent --no-tiling --node-exit $ne --refill-min $rf; done; done
run --workload coherent --static
run --workload incoherent --static
run --workload incoherent --static --node-exit 64
