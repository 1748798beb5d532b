# usage: profile_workload.sh <workload> <outdir-name>   -> gpurun_out/<outdir-name>/{trace,pmc_*}
# rocprofv3 evidence for the traversal kernel: kernel-trace stats, then PMC passes (separate runs;
# never combined with other trace domains).
WL=$1; OUT=$2
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --workload $WL"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/trace --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/$OUT/bench_trace.json 2> $R/gpurun_out/$OUT/bench_trace.err; rc=$?; echo "trace rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/$OUT/bench_trace.err; exit $rc; fi
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM" "sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE"; do
  set -- $pass; name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" -d $R/gpurun_out/$OUT/pmc_$name --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/$OUT/bench_$name.json 2> $R/gpurun_out/$OUT/bench_$name.err; rc=$?; echo "$name rc=$rc"
done
du -sh $R/gpurun_out/$OUT
