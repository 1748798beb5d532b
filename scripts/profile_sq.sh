# quick SQ counter pass for one bench invocation: profile_sq.sh <outdir> <bench args...>
OUT=$1; shift
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads $*"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY -d $R/gpurun_out/$OUT/pmc_sq --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/$OUT/bench_sq.json 2> $R/gpurun_out/$OUT/bench_sq.err; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH -d $R/gpurun_out/$OUT/pmc_sq2 --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/$OUT/bench_sq2.json 2> $R/gpurun_out/$OUT/bench_sq2.err; echo "sq2 rc=$?"
python3 - <<PY
import csv,glob,collections
for d in ["pmc_sq","pmc_sq2"]:
    for f in glob.glob("$R/gpurun_out/$OUT/"+d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if ("rtk_trace" in r["Kernel_Name"] or "rtk_packet_hot" in r["Kernel_Name"]) and "true" not in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "%.4g"%(sum(v)/len(v)))
PY
