#!/bin/bash
# On the GPU box: per-dispatch durations of k_recon (one launch per dependency level) and k_parse from a rocprofv3
# kernel trace of tools/kbench.py.   tools/recon_levels.sh
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_levels
timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_levels -- python $ROOT/tools/kbench.py 64 120 3 > $ROOT/gpurun_out/levels_kbench.log 2>&1
cd $ROOT
tail -1 gpurun_out/levels_kbench.log
python - <<PY
import sqlite3, glob
d = sorted(glob.glob('gpurun_out/prof_levels/**/*.db', recursive=True))[-1]
c = sqlite3.connect(d)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
k = [t for t in tabs if 'kernel' in t.lower()]
print(k)
rows = list(c.execute("select name, start, end from kernels order by start"))
recon = [(e - s) / 1e3 for n, s, e in rows if n.startswith('k_recon')]
parse = [(e - s) / 1e3 for n, s, e in rows if n.startswith('k_parse')]
print('k_parse us:', ['%.0f' % x for x in parse])
per = len(recon) // max(1, len(parse))
for i in range(0, len(recon), per):
    print('k_recon us by level:', ' '.join('%.0f' % x for x in recon[i:i + per]))
PY
