#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (all)"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
echo "== bench"; ( time timeout 900 python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err ) 2>&1 | tail -4; tail -25 gpurun_out/r04_bench_a.err; python -c "
import json
d=json.load(open('gpurun_out/r04_bench_a.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['reconstruct'], d['roofline']['ceiling_frac'])
for e in d.get('other_configs', []): print(e)
print(d['cpu_baseline'])
"
} > gpurun_out/r04_run5.txt 2>&1
tail -70 gpurun_out/r04_run5.txt
