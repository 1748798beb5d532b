#!/bin/bash
# round 4: rocprofv3 evidence for one workload: kernel-trace stats + PMC passes, summarised into gpurun_out/profiles_r04/
# usage: i_profile.sh <workload> [passes]
WL=$1

bash scripts/profile_workload.sh $WL prof_r04_$WL
case $WL in
  coherent) K="rtk_packet_beam2";;
  incoherent) K="rtk_lane_hot_closest";;
  shadow) K="rtk_lane_hot_any";;
esac
mkdir -p gpurun_out/profiles_r04
RTK_PROFILE_KERNEL="$K" RTK_PROFILE_WORKLOAD=$WL python3 scripts/summarize_profile.py gpurun_out/prof_r04_$WL gpurun_out/profiles_r04/r04_${WL}_lbvh > /dev/null
cat gpurun_out/profiles_r04/r04_${WL}_lbvh_kernel_stats.csv | head -12
