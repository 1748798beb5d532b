#!/bin/bash
# round 4: fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and duration of every kernel of the device build
# usage: k_build_traffic.sh <triangles> <tag>
N=${1:-10000000}; TAG=${2:-r4k}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/$TAG; mkdir -p $R/gpurun_out/$TAG
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/$TAG/$name --output-format csv -- python3 $R/scripts/build_timing.py $N > $R/gpurun_out/$TAG/$name.log 2>&1 || { echo "$name failed"; tail -3 $R/gpurun_out/$TAG/$name.log; exit 1; }
done
N=$N TAG=$TAG python3 - <<'PY'
import csv, glob, os, collections, re
R = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/" + os.environ["TAG"]
def short(name):
    m = re.search(r"(k_[a-z_0-9]+|rtk_[a-z_0-9]+|__amd_[a-zA-Z_]+)", name)
    return m.group(1) if m else name[:28]
n = int(os.environ["N"])
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        tot[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(R + "/fetch/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)): dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
builds = max(1, len(dur.get("k_morton", [1])))
print("# %d triangles, %d builds; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 half-count of 16-B reads, MI355X_MICROARCH.md); us and MB are per BUILD (all launches of the kernel in one build)" % (n, builds))
print("%-28s %9s %8s %10s %10s %10s %9s" % ("kernel", "us/build", "launches", "fetch MB", "write MB", "B/triangle", "GB/s"))
rows = []
for k, d in tot.items():
    calls = len(dur.get(k, []))
    if not calls: continue
    us = sum(dur[k]) / 1e3 / builds
    f = 2.0 * sum(d.get("FETCH_SIZE", [0])) * 1024 / builds
    w = sum(d.get("WRITE_SIZE", [0])) * 1024 / builds
    rows.append((us, k, calls / builds, f, w))
tf = tw = tu = 0.0
for us, k, calls, f, w in sorted(rows, reverse=True):
    tf += f; tw += w; tu += us
    if us < 3: continue
    print("%-28s %9.1f %8.1f %10.1f %10.1f %10.1f %9.0f" % (k, us, calls, f / 1e6, w / 1e6, (f + w) / n, (f + w) / (us * 1e-6) / 1e9 if us else 0))
print("%-28s %9.1f %8s %10.1f %10.1f %10.1f %9.0f" % ("all kernels", tu, "", tf / 1e6, tw / 1e6, (tf + tw) / n, (tf + tw) / (tu * 1e-6) / 1e9))
PY
