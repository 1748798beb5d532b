#!/bin/bash
# round 3: counters of the ray-pool kernel on the incoherent batch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3h
rm -rf $R/gpurun_out/r3h/*
for pass in "sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "sq2 SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "sq3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "tcc TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/r3h/$name --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads --workload ${WL:-incoherent} $EXTRA > $R/gpurun_out/r3h/$name.json 2> $R/gpurun_out/r3h/$name.err || { echo "$name failed"; tail -3 $R/gpurun_out/r3h/$name.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3h"
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rtk_trace" in row["Kernel_Name"]:
            tot[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kname, d in tot.items():
    print(kname)
    for k, x in sorted(d.items()): print("   %-24s %.4g (x%d)" % (k, sum(x) / len(x), len(x)))
for f in glob.glob(R + "/sq/**/*kernel_trace.csv", recursive=True):
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)): dur[row["Kernel_Name"][:60]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, x in dur.items():
        if "rtk_trace" in k: print("   duration", k, sum(x) / len(x) / 1e6, "ms")
PY
