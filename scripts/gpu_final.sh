# final evidence of the round: full GPU suite, the three bench lines (with the CPU baseline), rocprofv3 summaries,
# build stage timings, per-ray latency. Everything lands in gpurun_out/ and is copied to profiles/ by hand.
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; rc=$?; tail -6 gpurun_out/pytest_final.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py > gpurun_out/r03_build_timing.log 2>&1; echo "timing rc=$?"; grep -v amdgpu.ids gpurun_out/r03_build_timing.log
timeout -k 10 200 python scripts/single_ray_latency.py > gpurun_out/r03_single_ray_latency.log 2>&1; echo "latency rc=$?"; grep -v amdgpu.ids gpurun_out/r03_single_ray_latency.log
prof() { wl=$1; kern=$2; RTK_PROFILE_KERNEL="$kern" bash scripts/profile_workload.sh $wl prof_final_$wl > gpurun_out/prof_final_$wl.log 2>&1; RTK_PROFILE_WORKLOAD=$wl RTK_PROFILE_KERNEL="$kern" python3 scripts/summarize_profile.py gpurun_out/prof_final_$wl gpurun_out/r03_${wl}_lbvh > gpurun_out/r03_${wl}.summary 2>&1; tail -1 gpurun_out/prof_final_$wl.log; cp gpurun_out/r03_${wl}_lbvh_pmc.json gpurun_out/r03_${wl}_lbvh_kernel_stats.csv profiles/; }
prof coherent "rtk_packet_hot"
prof incoherent "rtk_trace_kernel<0, false, false, true>"
prof shadow "rtk_trace_kernel<1, false, false, true>"
for n in 1000000 10000000; do bash scripts/profile_build.sh $n > gpurun_out/r03_build_profile_$n.log 2>&1; cp gpurun_out/build_kernel_stats_$n.csv gpurun_out/r03_build_kernel_stats_$((n / 1000000))M.csv; grep -E "k_|rc=" gpurun_out/r03_build_profile_$n.log | head -12; done
gcc -O2 -o examples/host_latency examples/host_latency.c -Iinclude -Lrtk_amd -lrtk_amd -lpthread -Wl,-rpath,$PWD/rtk_amd 2>/dev/null && timeout -k 10 200 ./examples/host_latency > gpurun_out/r03_c_host_latency.log 2>&1; echo "c host rc=$?"; tail -6 gpurun_out/r03_c_host_latency.log
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench default rc=$?"
for wl in coherent incoherent shadow; do
  timeout -k 10 400 python bench.py --no-other-workloads --workload $wl > gpurun_out/r03_bench_$wl.json 2> gpurun_out/r03_bench_$wl.err; echo "bench $wl rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/r03_bench_$wl.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d.get('cpu_baseline',{})
print('$wl', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'limiter', r['limiter'] and {k:v for k,v in r['limiter'].items() if k!='note'}, 'build', d.get('build',{}).get('ms'))
print('   cpu', c.get('value'), c.get('cores'), {k: (v if k!='mismatching_rays' else len(v)) for k,v in c.get('parity_vs_gpu_oracle_bvh',{}).items()}, c.get('parity_vs_gpu_same_bvh',{}).get('ids_exact'))"
done
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_final.log
