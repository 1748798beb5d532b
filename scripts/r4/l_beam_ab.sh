# round 4: beam kernel A/B on the headline batch: RTK_AMD_PACKET_BEAM=0/1, 20/5 and 100/20
for b in 0 1 2 2; do
  RTK_AMD_PACKET_BEAM=$b timeout -k 10 300 python bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 100 --warmup 20 > gpurun_out/l_beam_$b.json 2> gpurun_out/l_beam_$b.err; echo "beam=$b rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/l_beam_$b.json').read().strip().splitlines()[-1])
print('beam=$b', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms'])"
done
