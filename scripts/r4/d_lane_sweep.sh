#!/bin/bash
# round 4: launch parameters of the assembly per-lane kernels re-swept (the C++ kernel's optimum was refill_min 8, node_exit 32)
mkdir -p gpurun_out
L=gpurun_out/r4d_sweep.log; : > $L
run() { timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', 'cells=%s' % os.environ.get('RTK_AMD_SORT_CELL_BITS','7'), d['value'], 'Mrays/s', d['roofline']['kernel_ms'])" >> $L || exit 1; }
for ne in 16 24 32 40 48; do for rm in 4 8 16 32; do
  run --workload incoherent --sort-rays --node-exit $ne --refill-min $rm
  run --workload shadow --node-exit $ne --refill-min $rm
done; done
for cb in 5 6 8; do RTK_AMD_SORT_CELL_BITS=$cb run --workload incoherent --sort-rays; RTK_AMD_SORT_CELL_BITS=$cb run --workload shadow; done
cat $L
