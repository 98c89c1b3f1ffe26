#!/bin/bash
# On the GPU box: GPU tests, the default bench line, and the three rocprofv3 passes (kernel trace + stats, FETCH_SIZE,
# WRITE_SIZE: separate passes, no trace domains beside --kernel-trace) whose summaries go to profiles/<tag>_*.
# (the profiled runs are the headline's passes alone, in ONE process: --no-napi --no-h2d -- the Node child's two batches in flight
# and the upload variant's passes would be averaged into the per-kernel figures)
#   tools/profile_round.sh r01c
tag="${1:-rXX}"
ROOT=$(pwd)
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q) > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -5 gpurun_out/${tag}_bench.err
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_trace $ROOT/gpurun_out/prof_fetch $ROOT/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_trace -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > $ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_rocprof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/prof_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/gpurun_out/prof_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-napi --no-h2d > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_write.err
rm -rf $ROOT/gpurun_out/prof_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $ROOT/gpurun_out/prof_valu -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-h2d --no-audio > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_valu.err
cd $ROOT
python tools/rocprof_summary.py $tag gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_valu > gpurun_out/${tag}_summary.txt 2>&1; tail -40 gpurun_out/${tag}_summary.txt
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/pmc_traffic.json profiles/pmc_valu.json gpurun_out/profiles_out/ 2>/dev/null
cat gpurun_out/${tag}_bench.json
