# final evidence of the round (round 4): full GPU suite, the bench lines (with the CPU baseline), rocprofv3 summaries of the three
# workloads, build kernel stats and traffic, per-ray latency, smoke. Everything lands in gpurun_out/; copied to profiles/ by hand.
R=r04
mkdir -p gpurun_out/profiles_$R
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; rc=$?; tail -4 gpurun_out/pytest_final.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py > gpurun_out/profiles_$R/${R}_build_timing.log 2>&1; echo "timing rc=$?"
PYTHONPATH=. timeout -k 10 200 python scripts/single_ray_latency.py > gpurun_out/profiles_$R/${R}_single_ray_latency.log 2>&1; echo "latency rc=$?"
timeout -k 10 120 ./examples/launch_latency_probe > gpurun_out/profiles_$R/${R}_launch_latency_probe.log 2>&1; echo "probe rc=$?"
gcc -O2 -o examples/host_latency examples/host_latency.c -Iinclude -Lrtk_amd -lrtk_amd -lpthread -Wl,-rpath,$PWD/rtk_amd 2>/dev/null && timeout -k 10 200 ./examples/host_latency > gpurun_out/profiles_$R/${R}_c_host_latency.log 2>&1; echo "c host rc=$?"
for wl in coherent incoherent shadow; do bash scripts/r4/i_profile.sh $wl > gpurun_out/prof_${R}_$wl.log 2>&1; echo "profile $wl rc=$?"; done
# (the bench lines below show `traffic` / `limiter` only from a summary taken on the SAME kernel code: use the ones just made)
cp gpurun_out/profiles_$R/${R}_*_pmc.json profiles/ 2>/dev/null
for n in 1000000 10000000; do bash scripts/profile_build.sh $n > gpurun_out/${R}_build_profile_$n.log 2>&1; cp gpurun_out/build_kernel_stats_$n.csv gpurun_out/profiles_$R/${R}_build_kernel_stats_$((n / 1000000))M.csv; done
bash scripts/r4/k_build_traffic.sh 10000000 r4k_10M > gpurun_out/profiles_$R/${R}_build_traffic.log 2>&1; bash scripts/r4/k_build_traffic.sh 1000000 r4k_1M >> gpurun_out/profiles_$R/${R}_build_traffic.log 2>&1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$R/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo "bench default rc=$?"
for wl in coherent incoherent shadow; do
  timeout -k 10 400 python bench.py --no-other-workloads --workload $wl > gpurun_out/profiles_$R/${R}_bench_$wl.json 2> gpurun_out/${R}_bench_$wl.err; echo "bench $wl rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/profiles_$R/${R}_bench_$wl.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d.get('cpu_baseline',{})
print('$wl', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'limiter', r['limiter'] and {k:v for k,v in r['limiter'].items() if k!='note'}, 'build', d.get('build',{}).get('ms'))
print('   cpu', c.get('value'), c.get('cores'), c.get('one_thread'), c.get('all_cores'), c.get('cgroup_cpu_quota_cores'), {k: (v if k!='mismatching_rays' else len(v)) for k,v in c.get('parity_vs_gpu_oracle_bvh',{}).items()}, c.get('parity_vs_gpu_same_bvh',{}).get('ids_exact'))"
done
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_final.log
