# round 5, first GPU call: the whole GPU suite on the morning's changes (per-ray host path, bench launcher, key rebuild, config-5 full size),
# the C host's per-ray latency, and the default bench line as the round's baseline
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r5/pytest_a.log 2>&1; rc=$?; tail -5 gpurun_out/r5/pytest_a.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
gcc -std=c11 -O2 -o examples/host_latency examples/host_latency.c -Iinclude -Lrtk_amd -lrtk_amd -lpthread -Wl,-rpath,$PWD/rtk_amd -Wl,-rpath,/opt/rocm/lib -lm && timeout -k 10 200 ./examples/host_latency 100000 200000 8 > gpurun_out/r5/c_host_latency.log 2>&1; echo "c host rc=$?"; cat gpurun_out/r5/c_host_latency.log
timeout -k 10 200 ./examples/host_latency 1000000 200000 8 >> gpurun_out/r5/c_host_latency.log 2>&1; tail -4 gpurun_out/r5/c_host_latency.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_a.json 2> gpurun_out/r5/bench_a.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['build'])
for k,v in d['other_workloads'].items(): print(k, v.get('value'), v.get('frac'), v.get('real_bound'), v.get('error'))"
