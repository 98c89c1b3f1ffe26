#!/bin/bash
# On the GPU box: average duration of every kernel of tools/kbench.py runs from a rocprofv3 kernel trace.
#   tools/kernel_times.sh [kbench args]
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_kt
timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_kt -- python $ROOT/tools/kbench.py "$@" > $ROOT/gpurun_out/kt_kbench.log 2>&1
cd $ROOT
tail -1 gpurun_out/kt_kbench.log
python - <<PY
import sqlite3, glob, collections
d = sorted(glob.glob('gpurun_out/prof_kt/**/*.db', recursive=True))[-1]
c = sqlite3.connect(d)
acc = collections.defaultdict(list)
for n, s, e in c.execute("select name, start, end from kernels order by start"):
    acc[n.split('(')[0]].append((e - s) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-28s calls %4d  avg %9.1f us  min %9.1f  max %9.1f' % (n[:28], len(v), sum(v) / len(v), min(v), max(v)))
PY
