# round 4: which packet kernel runs, how long, and its instruction counts (beam = $1)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B=${1:-1}
export RTK_AMD_PACKET_BEAM=$B
rm -rf gpurun_out/l_beam_prof_$B; mkdir -p gpurun_out/l_beam_prof_$B
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/l_beam_prof_$B/kt -o kt -- python3 bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 5 --warmup 2 > gpurun_out/l_beam_prof_$B/kt.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/l_beam_prof_$B/kt/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/l_beam_prof_$B/pmc -o pmc --output-format csv -- python3 bench.py --no-other-workloads --no-cpu-baseline --workload coherent --steps 3 --warmup 1 > gpurun_out/l_beam_prof_$B/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/l_beam_prof_$B/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rtk_packet' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
