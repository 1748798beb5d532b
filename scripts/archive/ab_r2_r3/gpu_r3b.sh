# round 3, B: instruction counters (SQ) of the packet kernel, new against round 2
mkdir -p gpurun_out
bash scripts/profile_sq.sh sq_r3b_new --workload coherent > gpurun_out/sq_r3b_new.log 2>&1
RTK_AMD_LIB=$PWD/build/libs/librtk_base.so bash scripts/profile_sq.sh sq_r3b_base --workload coherent > gpurun_out/sq_r3b_base.log 2>&1
echo new; grep -v amdgpu.ids gpurun_out/sq_r3b_new.log | tail -20; echo base; grep -v amdgpu.ids gpurun_out/sq_r3b_base.log | tail -20
