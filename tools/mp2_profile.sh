#!/bin/bash
# On the GPU box: MP2 stage GPU tests, smoke, its bench figure, a rocprofv3 kernel trace of it, then the default
# bench.py line (with audio_stage) and the whole GPU suite, in that order of priority (each under its own timeout);
# everything lands in gpurun_out/<tag>_*.
#   tools/mp2_profile.sh r01f
tag="${1:-rXX}"
ROOT=$(pwd)
mkdir -p gpurun_out
(time timeout 240 python -m pytest tests/test_mp2_gpu.py tests/test_mp2_node_host.py -m gpu -q) > gpurun_out/${tag}_mp2_pytest.log 2>&1; tail -6 gpurun_out/${tag}_mp2_pytest.log
(time timeout 200 python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/${tag}_smoke.log 2>&1; tail -3 gpurun_out/${tag}_smoke.log
timeout 120 python tools/mp2_bench.py > gpurun_out/${tag}_mp2_bench.json 2> gpurun_out/${tag}_mp2_bench.err || tail -5 gpurun_out/${tag}_mp2_bench.err
cat gpurun_out/${tag}_mp2_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_mp2
timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_mp2 -- python $ROOT/tools/mp2_bench.py --reps 20 > $ROOT/gpurun_out/${tag}_mp2_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_mp2_rocprof.err
cd $ROOT
find gpurun_out/prof_mp2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_mp2_kernel_stats.csv \;
head -12 gpurun_out/${tag}_mp2_kernel_stats.csv
find gpurun_out/prof_mp2 -name "*kernel_trace.csv" -size +8M -delete
(time timeout 420 python bench.py) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -5 gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_bench.json
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/${tag}_pytest_all.log 2>&1; tail -4 gpurun_out/${tag}_pytest_all.log
