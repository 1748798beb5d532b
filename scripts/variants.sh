for v in t4 t8 t16 t32; do
  cp variants/librtk_$v.so rtk_amd/librtk_amd.so
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload coherent 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], 'Mrays/s', d['config']['hit_fraction'])" || exit 1
done
cp variants/librtk_t8.so rtk_amd/librtk_amd.so
for b in 2 3 4; do timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload coherent --blocks-per-cu $b 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blocks_per_cu $b', d['value'], 'Mrays/s')" || exit 1; done
