#!/bin/bash
# On the GPU box: the three rocprofv3 passes of bench.py (kernel trace + stats, FETCH_SIZE, WRITE_SIZE; separate passes)
# and their summary -> gpurun_out/<tag>_summary.txt, profiles/<tag>_kernel_stats.txt, <tag>_pmc.txt, pmc_traffic.json
# (tools/profile_round.sh without the test and bench runs).     tools/profile_video.sh r01o
tag="${1:-rXX}"
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_trace $ROOT/gpurun_out/prof_fetch $ROOT/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_trace -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-audio > $ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_rocprof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/prof_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-audio > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/gpurun_out/prof_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-audio > /dev/null 2> $ROOT/gpurun_out/${tag}_rocprof_write.err
cd $ROOT
python tools/rocprof_summary.py $tag gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/${tag}_summary.txt 2>&1; tail -30 gpurun_out/${tag}_summary.txt
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/pmc_traffic.json gpurun_out/profiles_out/ 2>/dev/null
