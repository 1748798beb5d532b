#!/bin/bash
# round 4: how many rays the assembly per-lane kernels hand back, per workload
mkdir -p gpurun_out
export RTK_AMD_LANE_STATS=1
for wl in incoherent shadow; do
  timeout -k 10 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --workload $wl 2>&1 | grep "handed back" | sort | uniq -c
done
