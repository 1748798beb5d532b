#!/bin/bash
# round 4: instruction counters of the per-lane kernels (assembly, and C++ for comparison), per 64 rays
# usage: c_lane_counters.sh <tag> [bench args...]
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM" "tcc TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass "$@"; name=$1; shift
  ctrs=""; while [ $# -gt 0 ] && [[ "$1" != --* ]]; do ctrs="$ctrs $1"; shift; done
  timeout -k 10 300 rocprofv3 --pmc $ctrs -d $R/gpurun_out/$TAG/pmc_$name --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > $R/gpurun_out/$TAG/bench_$name.json 2> $R/gpurun_out/$TAG/bench_$name.err; echo "$name rc=$?"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/$TAG/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "rtk_lane_hot" in k or ("rtk_trace_kernel" in k):
            agg[(k[:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
per64 = (1<<24)/64.0
for k,v in sorted(agg.items()):
    m=sum(v)/len(v)
    print("%-42s %-24s %12.4g  per 64 rays %10.1f" % (k[0], k[1], m, m/per64))
PY
