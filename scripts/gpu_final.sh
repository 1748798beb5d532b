# final evidence of a round (R below): full GPU suite, the bench lines (with the CPU baseline), rocprofv3 summaries of the three
# workloads (kernel trace + PMC passes, tied to the kernels' machine code), build kernel stats and traffic, per-ray latency, the
# cpu-sah A/B, smoke. Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand afterwards.
R=${R:-r05}
mkdir -p gpurun_out/profiles_$R
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; rc=$?; tail -4 gpurun_out/pytest_final.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/build_timing.py > gpurun_out/profiles_$R/${R}_build_timing.log 2>&1; echo "timing rc=$?"
gcc -std=c11 -O2 -o examples/host_latency examples/host_latency.c -Iinclude -Lrtk_amd -lrtk_amd -lpthread -lm -Wl,-rpath,$PWD/rtk_amd -Wl,-rpath,/opt/rocm/lib 2>/dev/null && { timeout -k 10 200 ./examples/host_latency 100000 200000 8; timeout -k 10 200 ./examples/host_latency 1000000 200000 8; echo "# RTK_AMD_PER_RAY=gpu (the one-ray kernel), 100000 triangles:"; RTK_AMD_PER_RAY=gpu timeout -k 10 200 ./examples/host_latency 100000 2000 4; } > gpurun_out/profiles_$R/${R}_c_host_latency.log 2>&1; echo "c host rc=$?"
for wl in coherent incoherent shadow; do
  bash scripts/profile_workload.sh $wl prof_${R}_$wl > gpurun_out/prof_${R}_$wl.log 2>&1
  case $wl in coherent) K="rtk_packet_beam2";; incoherent) K="rtk_lane_hot_closest";; shadow) K="rtk_lane_hot_any";; esac
  RTK_PROFILE_KERNEL="$K" RTK_PROFILE_WORKLOAD=$wl python3 scripts/summarize_profile.py gpurun_out/prof_${R}_$wl gpurun_out/profiles_$R/${R}_${wl}_lbvh > /dev/null; echo "profile $wl rc=$?"
done
# (the bench lines below show `traffic` / `limiter` only from a summary taken on the SAME kernel code: use the ones just made)
cp gpurun_out/profiles_$R/${R}_*_pmc.json profiles/ 2>/dev/null
for n in 1000000 10000000; do bash scripts/profile_build.sh $n > gpurun_out/${R}_build_profile_$n.log 2>&1; cp gpurun_out/build_kernel_stats_$n.csv gpurun_out/profiles_$R/${R}_build_kernel_stats_$((n / 1000000))M.csv; done
bash scripts/r4/k_build_traffic.sh 10000000 r5k_10M > gpurun_out/profiles_$R/${R}_build_traffic.log 2>&1; bash scripts/r4/k_build_traffic.sh 1000000 r5k_1M >> gpurun_out/profiles_$R/${R}_build_traffic.log 2>&1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$R/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo "bench default rc=$?"
for wl in coherent incoherent shadow; do
  timeout -k 10 400 python bench.py --no-other-workloads --workload $wl > gpurun_out/profiles_$R/${R}_bench_$wl.json 2> gpurun_out/${R}_bench_$wl.err; echo "bench $wl rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/profiles_$R/${R}_bench_$wl.json').read().strip().splitlines()[-1])
r=d['roofline']; c=d.get('cpu_baseline',{})
print('$wl', d['value'], 'Mrays/s kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'real_bound', r.get('real_bound'), 'build', d.get('build',{}).get('ms'), 'no hint', d['config'].get('without_image_hint'))
print('   cpu', c.get('value'), c.get('cores'), c.get('one_thread'), c.get('all_cores'), c.get('cgroup_cpu_quota_cores'), {k: (v if k!='mismatching_rays' else len(v)) for k,v in c.get('parity_vs_gpu_oracle_bvh',{}).items()}, c.get('parity_vs_gpu_same_bvh',{}).get('ids_exact'))"
done
# the reference's leaf sizes (CPU task builder's SAH tree, leaves of 4 .. 63) on the hand-written kernels and on the C++ ones
{ for wl in coherent incoherent; do for asm in 1 0; do RTK_AMD_PACKET_ASM=$asm RTK_AMD_LANE_ASM=$asm timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --workload $wl --bvh cpu-sah 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl cpu-sah tree, hand-written kernels %s:' % ('on' if $asm else 'off (C++ kernels)'), d['value'], 'Mrays/s, kernel', d['roofline']['kernel_ms'], 'ms, visits per ray', d['roofline']['visits_per_ray'])"; done; done; } > gpurun_out/profiles_$R/${R}_cpu_sah_ab.log 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_final.log
