# round 4: why the packet kernels wait -- average latency of vector / scalar loads (SQ_INST_LEVEL_* / SQ_INSTS_*), instruction fetch
# stalls, cache hit rates. $1 = RTK_AMD_PACKET_BEAM (0 / 1)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B=${1:-1}
export RTK_AMD_PACKET_BEAM=$B
D=gpurun_out/l_beam_pmc_$B; rm -rf $D; mkdir -p $D
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_IFETCH SQ_IFETCH" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $D/p$i -o pmc --output-format csv -- python3 bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 3 --warmup 1 > $D/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$D/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('rtk_packet'):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in acc.items()}
for c in sorted(m): print('%-32s %16.0f' % (c, m[c]))
def g(k): return m.get(k, float('nan'))
print('avg VMEM latency (cycles)', g('SQ_INST_LEVEL_VMEM') / g('SQ_INSTS_VMEM_RD'))
print('avg SMEM latency (cycles)', g('SQ_INST_LEVEL_SMEM') / g('SQ_INSTS_SMEM'))
print('wave cycles per tile', g('SQ_WAVE_CYCLES') / 262144 * 4 if False else g('SQ_WAVE_CYCLES') / 262144)
PY
