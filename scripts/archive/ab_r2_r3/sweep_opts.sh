for wl in coherent incoherent; do
for ne in 1 8 16 24 32 48 64; do
 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl --node-exit $ne 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl node_exit=$ne', d['value'], 'Mrays/s')" || exit 1
done
for rf in 8 16 32 48; do
 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl --refill-min $rf 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl refill_min=$rf', d['value'], 'Mrays/s')" || exit 1
done
done
