# per-kernel times of the device build (rocprofv3 kernel trace of scripts/build_timing.py for one size)
N=${1:-10000000}
mkdir -p gpurun_out/prof_build_$N
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_build_$N --output-format csv -- python3 $R/scripts/build_timing.py $N > $R/gpurun_out/prof_build_$N/out.log 2>&1
echo "rc=$?"
f=$(find $R/gpurun_out/prof_build_$N -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print('%-70s calls %5s avg_us %9.1f total_us %10.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
cp "$f" $R/gpurun_out/build_kernel_stats_$N.csv
