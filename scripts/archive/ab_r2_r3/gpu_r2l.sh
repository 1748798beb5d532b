run() { lib=$1; shift; RTK_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('lib=%s %s' % ('$lib'.split('/')[-1] or 'default', '$*'), d['value'], 'Mrays/s kernel_ms', r['kernel_ms'])" || echo "FAILED $lib $*"; }
for rep in 1 2; do for lib in "" $PWD/build/libs/librtk_pkA.so $PWD/build/libs/librtk_pkB.so; do run "$lib" --workload coherent; done; done
