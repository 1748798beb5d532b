# round 4: entry-list parameters under the beam kernel (RTK_AMD_ENTRY_TARGET x RTK_AMD_ENTRY_LEVELS), headline batch
for cfg in "26 8" "12 6" "40 8" "52 10" "26 10" "0 0"; do
  set -- $cfg
  if [ "$1" = "0" ]; then export RTK_AMD_PACKET_ENTRIES=0; else export RTK_AMD_PACKET_ENTRIES=1 RTK_AMD_ENTRY_TARGET=$1 RTK_AMD_ENTRY_LEVELS=$2; fi
  timeout -k 10 300 python bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 50 --warmup 10 > gpurun_out/l_ent.json 2> gpurun_out/l_ent.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/l_ent.json').read().strip().splitlines()[-1])
print('target $1 levels $2:', d['value'], 'Mrays/s ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
