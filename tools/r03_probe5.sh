#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "== bench --force-dist (one rank: both ingest modes)"; timeout 600 python bench.py --force-dist --steps 4 --warmup 1 --no-cpu-baseline --no-audio 2> gpurun_out/r03_force_dist.err > gpurun_out/r03_force_dist.json; tail -8 gpurun_out/r03_force_dist.err; cat gpurun_out/r03_force_dist.json
echo "== bench --gpus 2 on this box"; timeout 300 python bench.py --gpus 2 --steps 2 2>&1 | tail -3
} > gpurun_out/r03_probe5.txt 2>&1
tail -60 gpurun_out/r03_probe5.txt
