# rocprofv3 evidence for the traversal kernel: kernel-trace stats, then PMC passes (separate runs).
set -o pipefail
mkdir -p gpurun_out/prof_r1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 5 --warmup 1 --no-cpu-baseline $BENCH_EXTRA"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1/trace --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r1/bench_trace.json 2> $R/gpurun_out/prof_r1/bench_trace.err; rc=$?; echo "trace rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r1/bench_trace.err; exit $rc; fi
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r1/pmc_fetch --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r1/bench_fetch.json 2> $R/gpurun_out/prof_r1/bench_fetch.err; rc=$?; echo "fetch rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r1/bench_fetch.err; exit $rc; fi
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r1/pmc_write --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r1/bench_write.json 2> $R/gpurun_out/prof_r1/bench_write.err; rc=$?; echo "write rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r1/bench_write.err; exit $rc; fi
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_r1/pmc_tcc --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r1/bench_tcc.json 2> $R/gpurun_out/prof_r1/bench_tcc.err; rc=$?; echo "tcc rc=$rc"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $R/gpurun_out/prof_r1/pmc_sq --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r1/bench_sq.json 2> $R/gpurun_out/prof_r1/bench_sq.err; rc=$?; echo "sq rc=$rc"
cd $R/gpurun_out/prof_r1 && find . -name "*.csv" | head -40 && du -sh .
