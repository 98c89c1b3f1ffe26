#!/bin/bash
# On the GPU box: tools/enc_content_bench.py --gops all (64 x 120 pictures of encoder-made 1080p content, every GOP of tools/enc_content.py: tests/enc/_cache/ must hold them; gated against the oracle) on the
# host clock, then under rocprofv3: kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU (a pass each) -> gpurun_out/r06k_*
ROOT=$(pwd)
mkdir -p gpurun_out
python tools/enc_content_bench.py 64 10 8 --gops all --out gpurun_out/r06k_enc_content_all_gops.json 2> gpurun_out/r06k_enc_content.err | cut -c1-600 || tail -5 gpurun_out/r06k_enc_content.err
# the headline's content on the same box, same clock (kbench: the engine's phase events)
python tools/kbench.py 64 120 8 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r06k_cfg2_same_box.txt; cat gpurun_out/r06k_cfg2_same_box.txt
cd /tmp; export TMPDIR=/tmp
for pass in trace fetch write valu; do
  rm -rf $ROOT/gpurun_out/prof_enc_$pass
  case $pass in
    trace) opt="--kernel-trace --stats";;
    fetch) opt="--kernel-trace --pmc FETCH_SIZE";;
    write) opt="--kernel-trace --pmc WRITE_SIZE";;
    valu)  opt="--kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES";;
  esac
  timeout 400 rocprofv3 $opt -d $ROOT/gpurun_out/prof_enc_$pass -- python $ROOT/tools/enc_content_bench.py 64 10 3 --gops all > /dev/null 2> $ROOT/gpurun_out/r06k_rocprof_$pass.err
done
cd $ROOT
python - > gpurun_out/r06k_enc_content_kernels.txt 2>&1 <<'P'
import glob, os, sqlite3
def db(d): return sqlite3.connect(sorted(glob.glob(os.path.join("gpurun_out", d, "**", "*.db"), recursive=True))[-1])
short = lambda n: n.split("(")[0]
print("# rocprofv3 --kernel-trace --stats -- python tools/enc_content_bench.py 64 10 3 --gops all   (5 decode passes; microseconds)")
print("%-28s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db("prof_enc_trace").execute("select * from top_kernels"):
    print("%-28s %8d %14.1f %12.2f %7.2f%%" % (short(name)[:28], calls, total, avg, pct))
q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name"
f = {short(r[0]): r[2] for r in db("prof_enc_fetch").execute(q, ("FETCH_SIZE",))}
w = {short(r[0]): r[2] for r in db("prof_enc_write").execute(q, ("WRITE_SIZE",))}
print("\n# --pmc FETCH_SIZE / --pmc WRITE_SIZE (a pass each), per dispatch; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950's FETCH_SIZE counts half)")
for k in sorted(set(f) | set(w)):
    if k.startswith("k_"): print("%-16s fetch %12.1f KiB  write %12.1f KiB  hbm %10.1f MB per launch" % (k, f.get(k, 0), w.get(k, 0), (2 * f.get(k, 0) + w.get(k, 0)) * 1024 / 1e6))
v = {short(r[0]): r[2] for r in db("prof_enc_valu").execute(q, ("SQ_INSTS_VALU",))}
s = {short(r[0]): r[2] for r in db("prof_enc_valu").execute(q, ("SQ_WAVES",))}
print("\n# --pmc SQ_INSTS_VALU SQ_WAVES: vector instructions per wavefront")
for k in sorted(v):
    if k.startswith("k_") and s.get(k): print("%-16s %14.0f instructions %10.0f wavefronts %8.1f per wavefront" % (k, v[k], s[k], v[k] / s[k]))
P
cat gpurun_out/r06k_enc_content_kernels.txt
