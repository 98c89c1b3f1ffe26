tag=r06l
ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_trace $ROOT/gpurun_out/prof_fetch $ROOT/gpurun_out/prof_write $ROOT/gpurun_out/prof_valu
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_trace -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > $ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_rocprof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/prof_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/gpurun_out/prof_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $ROOT/gpurun_out/prof_valu -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d --no-audio > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_valu.err
cd $ROOT
python tools/rocprof_summary.py $tag gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_valu > gpurun_out/${tag}_summary.txt 2>&1; head -30 gpurun_out/${tag}_summary.txt
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/pmc_traffic.json profiles/pmc_valu.json gpurun_out/profiles_out/
ls gpurun_out/prof_trace/*/
