mkdir -p gpurun_out
echo skip-tests; rc=0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof_build
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_build/trace --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload shadow > $R/gpurun_out/prof_build/bench.json 2> $R/gpurun_out/prof_build/bench.err; echo "rc=$?"
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/prof_build/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['bvh_build_ms_in_library'], d['config']['bvh_build_mtris_s'])"
cut -c1-110 $R/gpurun_out/prof_build/trace/*/*_kernel_stats.csv | head -16
