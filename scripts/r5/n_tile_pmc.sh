# round 5: what k_collapse_tile waits for -- counters per launch (10M-triangle build)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
D=gpurun_out/r5/tile_pmc; rm -rf $D; mkdir -p $D
rocprofv3 --list-avail > $D/avail.txt 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $D/p$i -o pmc --output-format csv -- python3 scripts/build_timing.py 10000000 > $D/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$D/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ('k_collapse_tile', 'k_refit_tile', 'k_emit_tris'):
            if k in r['Kernel_Name']:
                acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for c in sorted(acc): print('%-18s %-32s %16.0f' % (c[0], c[1], sum(acc[c]) / len(acc[c])))
PY
