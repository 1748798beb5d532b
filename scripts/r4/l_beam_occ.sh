# round 4: the beam kernel at 2, 4, 6, 8 resident workgroups per CU (waves per SIMD): latency-bound or throughput-bound?
for b in 7 6; do
  RTK_AMD_HOT_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 30 --warmup 5 > gpurun_out/l_occ.json 2> gpurun_out/l_occ.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/l_occ.json').read().strip().splitlines()[-1])
print('workgroups per CU $b:', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'])"
done
