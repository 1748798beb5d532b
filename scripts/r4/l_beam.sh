# round 4: rtk_packet_beam against rtk_packet_hot on the headline batch (config 2), after the packet tests
set -x
timeout -k 10 600 python -m pytest tests/test_gpu_lane_asm.py tests/test_gpu_trace.py -m gpu -q -x > gpurun_out/l_beam_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/l_beam_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for b in 0 1; do
  RTK_AMD_PACKET_BEAM=$b timeout -k 10 300 python bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 20 --warmup 5 > gpurun_out/l_beam_$b.json 2> gpurun_out/l_beam_$b.err; echo "beam=$b rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/l_beam_$b.json').read().strip().splitlines()[-1])
print('beam=$b', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'])"
done
