# kernel-trace stats of one bench invocation: gpu_trace_stats.sh <outdir> <bench args...>
OUT=$1; shift
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads "$@" > $R/gpurun_out/$OUT/bench.json 2> $R/gpurun_out/$OUT/bench.err; echo "trace rc=$?"
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/$OUT/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
PY
