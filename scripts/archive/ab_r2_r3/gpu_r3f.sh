#!/bin/bash
# round 3: counters of the per-lane kernel on the incoherent batch, as given and re-ordered (does the miss rate or the vector unit bind?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3f
for v in given sorted; do
  EXTRA=""; [ $v = sorted ] && EXTRA="--sort-rays"
  for pass in "tcc TCC_HIT_sum TCC_MISS_sum" "sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "sq2 SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    set -- $pass; name=$1; shift
    RTK_AMD_SORT_CELL_BITS=6 timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/r3f/${v}_$name --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads --workload incoherent $EXTRA > $R/gpurun_out/r3f/${v}_$name.json 2> $R/gpurun_out/r3f/${v}_$name.err || { echo "$v $name failed"; tail -3 $R/gpurun_out/r3f/${v}_$name.err; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3f"
for v in ("given", "sorted"):
    tot = collections.defaultdict(list)
    for f in glob.glob(R + "/%s_*/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "rtk_trace_kernel" in row["Kernel_Name"]:
                tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(v, {k: "%.4g (x%d)" % (sum(x) / len(x), len(x)) for k, x in sorted(tot.items())})
PY
