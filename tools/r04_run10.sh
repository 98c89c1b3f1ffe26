#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (all)"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
echo "== bench --force-dist"; ( time timeout 900 python bench.py --force-dist --steps 5 --no-cpu-baseline --no-audio > gpurun_out/r04_bench_fd.json 2> gpurun_out/r04_bench_fd.err ) 2>&1 | tail -4; tail -12 gpurun_out/r04_bench_fd.err; python -c "
import json
d=json.load(open('gpurun_out/r04_bench_fd.json'))
print(d['value'], d['ms_per_step'], d['exchange'])
"
} > gpurun_out/r04_run10.txt 2>&1
tail -40 gpurun_out/r04_run10.txt
