# one PMC pass for one bench invocation: profile_pmc.sh <outdir> "<counters>" <bench args...>
OUT=$1; CTRS=$2; shift; shift
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc $CTRS -d $R/gpurun_out/$OUT/pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > $R/gpurun_out/$OUT/bench.json 2> $R/gpurun_out/$OUT/bench.err; echo "pmc rc=$?"
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/$OUT/pmc/*/*_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("rtk_trace" in r["Kernel_Name"] or "rtk_packet" in r["Kernel_Name"]) and "true" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:30],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(k, "%.4g"%(sum(v)/len(v)))
PY
