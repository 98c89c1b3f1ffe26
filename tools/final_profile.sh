#!/bin/bash
# On the GPU box: the whole GPU suite, the default bench line, the MP2 stage figure + rocprofv3 kernel trace and
# FETCH_SIZE / WRITE_SIZE passes of it, the latency probe -> gpurun_out/<tag>_*.
#   tools/final_profile.sh r01i
tag="${1:-rXX}"
ROOT=$(pwd)
mkdir -p gpurun_out
(time timeout 600 python -m pytest tests -m gpu -q) > gpurun_out/${tag}_pytest.log 2>&1; tail -4 gpurun_out/${tag}_pytest.log
(time timeout 420 python bench.py) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -5 gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_bench.json | cut -c1-400
timeout 120 python tools/mp2_bench.py --reps 9 > gpurun_out/${tag}_mp2_bench.json 2> gpurun_out/${tag}_mp2_bench.err || tail -5 gpurun_out/${tag}_mp2_bench.err
cat gpurun_out/${tag}_mp2_bench.json
timeout 200 python tools/latency_probe.py > gpurun_out/${tag}_latency.txt 2>&1; cat gpurun_out/${tag}_latency.txt
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_mp2 $ROOT/gpurun_out/prof_mp2_fetch $ROOT/gpurun_out/prof_mp2_write
timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_mp2 -- python $ROOT/tools/mp2_bench.py --reps 20 > $ROOT/gpurun_out/${tag}_mp2_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_mp2_rocprof.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/prof_mp2_fetch -- python $ROOT/tools/mp2_bench.py --reps 4 > /dev/null 2> $ROOT/gpurun_out/${tag}_mp2_rocprof_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/gpurun_out/prof_mp2_write -- python $ROOT/tools/mp2_bench.py --reps 4 > /dev/null 2> $ROOT/gpurun_out/${tag}_mp2_rocprof_write.err
cd $ROOT
python tools/mp2_rocprof_summary.py $tag gpurun_out/prof_mp2 gpurun_out/prof_mp2_fetch gpurun_out/prof_mp2_write > gpurun_out/${tag}_mp2_summary.txt 2>&1; cat gpurun_out/${tag}_mp2_summary.txt
