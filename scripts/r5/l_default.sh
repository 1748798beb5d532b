mkdir -p gpurun_out/r5
t0=$(date +%s)
python bench.py > gpurun_out/r5/bench_noflags.json 2> gpurun_out/r5/bench_noflags.err; echo "rc=$? seconds=$(( $(date +%s) - t0 ))"
python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_noflags.json').read().strip().splitlines()[-1])
print(d['value'], d['steps'], d['warmup'], d['roofline']['frac'], d['cpu_baseline']['value'], list(d['other_workloads'].keys()))"
