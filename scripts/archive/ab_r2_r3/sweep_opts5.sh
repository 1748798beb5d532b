run() { timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], 'Mrays/s')" || exit 1; }
for ne in 16 24 32; do for rf in 4 8 16; do run --workload incoherent --node-exit $ne --refill-min $rf; done; done
for ne in 16 24 32; do for rf in 8 16; do run --workload shadow --node-exit $ne --refill-min $rf; done; done
