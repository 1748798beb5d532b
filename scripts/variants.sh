for v in w5 w6 w8; do
  cp variants/librtk_$v.so rtk_amd/librtk_amd.so
  timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload coherent 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], 'Mrays/s', d['config']['hit_fraction'])" || exit 1
done
