#!/bin/bash
# round 3: instruction counters of the 10M-triangle build's kernels (which of them are bound by their arithmetic?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r3n; mkdir -p $R/gpurun_out/r3n
for pass in "sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "sq2 SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  set -- $pass; name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/r3n/$name --output-format csv -- python3 $R/scripts/build_timing.py 10000000 > $R/gpurun_out/r3n/$name.log 2>&1 || { echo "$name failed"; tail -3 $R/gpurun_out/r3n/$name.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3n"
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("::")[-1][:28]
        tot[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(R + "/sq/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)): dur[row["Kernel_Name"].split("(")[0].split("::")[-1][:28]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, d in sorted(tot.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    m = lambda c: sum(d[c]) / len(d[c]) if c in d else 0.0
    ns = sum(dur[k]) / len(dur[k]) if k in dur else 0
    if ns < 20000: continue
    clock = m("SQ_BUSY_CYCLES") / 32 / ns if ns else 0
    print("%-28s %7.1f us x%3d  valu/launch %.3g  valu busy %.2f  lane use %.2f  waiting %.2f  clock %.2f" % (k, ns / 1e3, len(dur[k]), m("SQ_INSTS_VALU"),
          m("SQ_ACTIVE_INST_VALU") * 4 / (1024 * ns * clock) if clock else 0, m("SQ_THREAD_CYCLES_VALU") / (64 * m("SQ_ACTIVE_INST_VALU")) if m("SQ_ACTIVE_INST_VALU") else 0,
          m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES") if m("SQ_WAVE_CYCLES") else 0, clock))
PY
