#!/bin/bash
# On the GPU box: the MP2 stage's GPU tests, its bench figure and a rocprofv3 kernel trace of it -> gpurun_out/<tag>_*.
#   tools/mp2_quick.sh r01g
tag="${1:-rXX}"
ROOT=$(pwd)
mkdir -p gpurun_out
(time timeout 240 python -m pytest tests/test_mp2_gpu.py tests/test_mp2_node_host.py -m gpu -q) > gpurun_out/${tag}_mp2_pytest.log 2>&1; tail -4 gpurun_out/${tag}_mp2_pytest.log
timeout 120 python tools/mp2_bench.py --reps 9 > gpurun_out/${tag}_mp2_bench.json 2> gpurun_out/${tag}_mp2_bench.err || tail -5 gpurun_out/${tag}_mp2_bench.err
cat gpurun_out/${tag}_mp2_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_mp2
timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_mp2 -- python $ROOT/tools/mp2_bench.py --reps 20 > $ROOT/gpurun_out/${tag}_mp2_bench_under_rocprof.json 2> $ROOT/gpurun_out/${tag}_mp2_rocprof.err
cd $ROOT
python - <<PY
import sqlite3, glob
d = sorted(glob.glob('gpurun_out/prof_mp2/**/*.db', recursive=True))[-1]
c = sqlite3.connect(d)
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    print("%-28s %8d %14.1f %12.2f %7.2f%%" % (name.split('(')[0][:28], calls, total, avg, pct))
PY
