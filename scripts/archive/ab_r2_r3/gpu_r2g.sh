# full GPU suite + rocprofv3 evidence for the three workloads (round 2 kernels)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2g.log 2>&1; rc=$?; tail -8 gpurun_out/pytest_r2g.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
prof() { wl=$1; kern=$2; RTK_PROFILE_KERNEL="$kern" bash scripts/profile_workload.sh $wl prof_r2g_$wl > gpurun_out/prof_r2g_$wl.log 2>&1; RTK_PROFILE_KERNEL="$kern" python3 scripts/summarize_profile.py gpurun_out/prof_r2g_$wl gpurun_out/r02_${wl}_lbvh > gpurun_out/r02_${wl}.summary 2>&1; tail -2 gpurun_out/prof_r2g_$wl.log; }
prof coherent "rtk_trace_packet_kernel<false>"
prof incoherent "rtk_trace_kernel<0, false, false, true>"
prof shadow "rtk_trace_kernel<1, false, false, true>"
ls -la gpurun_out/r02_*
