#!/bin/bash
# round 4: kernel-trace stats of one bench invocation: j_trace.sh <tag> [bench args]
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err; echo "trace rc=$?"
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/$TAG/trace/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
