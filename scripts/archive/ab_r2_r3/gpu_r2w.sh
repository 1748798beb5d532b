# packet kernel: waves per SIMD the register allocator has to leave room for (PK_MIN_WAVES), re-swept after the control-flow trims
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG $*', d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" || exit 1; }
mkdir -p gpurun_out
{
TAG=6waves run --workload coherent
for w in 5 7 8; do export RTK_AMD_LIB=$PWD/build/libs/librtk_pk$w.so; TAG=${w}waves run --workload coherent; done
} 2>&1 | tee gpurun_out/ab_r2w.log
