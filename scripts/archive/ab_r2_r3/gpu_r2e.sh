# compare exact vs compressed nodes under the profiler (incoherent rays)
mkdir -p gpurun_out
for qn in 0 1; do
  export RTK_AMD_QNODES=$qn
  RTK_PROFILE_KERNEL="rtk_trace_kernel<0, false, false, $( [ $qn = 1 ] && echo true || echo false )>" bash scripts/profile_workload.sh incoherent prof_r2e_q$qn > gpurun_out/prof_r2e_q$qn.log 2>&1
  RTK_PROFILE_KERNEL="rtk_trace_kernel<0, false, false, $( [ $qn = 1 ] && echo true || echo false )>" python3 scripts/summarize_profile.py gpurun_out/prof_r2e_q$qn gpurun_out/r2e_q$qn > gpurun_out/r2e_q$qn.summary 2>&1
  tail -3 gpurun_out/prof_r2e_q$qn.log
done
python3 - <<'PY'
import json
for q in (0,1):
    d=json.load(open('gpurun_out/r2e_q%d_pmc.json'%q))
    c=d['pmc_per_launch']
    print('qnodes',q,'avg_ns',d.get('kernel_trace'),'traffic',d.get('hbm_traffic_bytes_per_launch'),'l2hit',d.get('l2_hit_rate'))
    print('   ', {k: round(v['mean']) for k,v in c.items()})
PY
