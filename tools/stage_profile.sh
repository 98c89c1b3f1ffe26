#!/bin/bash
# rocprofv3 --kernel-trace --stats of the ingest (k_ts_*) and renderer (k_rgba) stages -> profiles/<tag>_stage_kernel_stats.txt
tag="${1:-rXX}"; ROOT=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_stage_ts $ROOT/gpurun_out/prof_stage_rgba
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_stage_ts -- python $ROOT/tools/ts_bench.py > $ROOT/gpurun_out/${tag}_ts_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_stage_rgba -- python $ROOT/tools/rgba_bench.py > $ROOT/gpurun_out/${tag}_rgba_bench.txt 2>&1
cd $ROOT
python - "$tag" <<'PY'
import glob, sqlite3, sys
tag = sys.argv[1]
out = open("gpurun_out/%s_stage_kernel_stats.txt" % tag, "w")
for name, d, cmd in (("ingest", "gpurun_out/prof_stage_ts", "tools/ts_bench.py (cfg2: 64 x 120-picture 1080p streams as MPEG-TS, 0.53 GB, 2.9 M packets)"),
                     ("renderer", "gpurun_out/prof_stage_rgba", "tools/rgba_bench.py (8 x 120 1080p pictures resident, 6 conversions of all 960)")):
    db = sqlite3.connect(sorted(glob.glob(d + "/**/*.db", recursive=True))[-1])
    out.write("# rocprofv3 --kernel-trace --stats -- python %s\n%-28s %8s %14s %12s\n" % (cmd, "kernel", "calls", "total_us", "avg_us"))
    for k, calls, total, avg, pct in db.execute("select * from top_kernels"):
        k = k.split("(")[0]
        if k.startswith("k_") or k.startswith("void k_"): out.write("%-28s %8d %14.1f %12.2f\n" % (k[:28], calls, total, avg))
    out.write("\n")
out.close()
print(open("gpurun_out/%s_stage_kernel_stats.txt" % tag).read())
PY
tail -2 gpurun_out/${tag}_ts_bench.txt; tail -1 gpurun_out/${tag}_rgba_bench.txt
